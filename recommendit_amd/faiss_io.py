"""Read / write FAISS ``IndexIVFFlat`` / ``IndexFlatIP`` files (what ``faiss.write_index`` / ``faiss.read_index``
exchange) without faiss -- SURVEY.md §8f-3: indexes built by either stack interoperate
(reference src/models/faiss_index.py:159-205: ``faiss.write_index(self.index, path)`` / ``faiss.read_index(path)``).

**Parity unpinned.**  faiss (``faiss-cpu>=1.7.4``, requirements.txt:2) is not installed here and the reference ships
no index file, so the byte layout below is restated from faiss 1.7.x's ``impl/index_write.cpp`` / ``index_read.cpp``
as published (little-endian, x86-64 type widths: int = 4, size_t = idx_t = 8, bool = 1) and is checked only by the
writer -> reader round trip and by hand-assembled byte strings in tests/test_host_logic.py:

  IndexFlatIP  : fourcc "IxFI" | header | size_t n_floats | float32[n_floats]                      (xb, row-major)
  IndexIVFFlat : fourcc "IwFl" | header | size_t nlist | size_t nprobe | <quantizer index> | direct map
                 | inverted lists
  header       : int d | idx_t ntotal | idx_t 1<<20 | idx_t 1<<20 | bool is_trained | int metric (0 = inner product,
                 1 = L2) [| float metric_arg if metric > 1]
  direct map   : char type (0 = none, 1 = array, 2 = hashtable) | size_t n | idx_t[n]  [| hashtable: size_t n | pairs]
  lists "ilar" : fourcc | size_t nlist | size_t code_size (= 4 d) | fourcc "full" + vector<size_t> sizes[nlist]
                 or fourcc "sprs" + vector<size_t> (list, size) pairs | per non-empty list: codes (n x code_size bytes)
                 then idx_t ids[n]
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import Dict, Optional

import numpy as np

METRIC_INNER_PRODUCT, METRIC_L2 = 0, 1


class FaissFormatError(ValueError):
    pass


class _Reader:
    def __init__(self, data: bytes):
        self.b, self.o = data, 0

    def take(self, n: int) -> bytes:
        if self.o + n > len(self.b):
            raise FaissFormatError("truncated FAISS index file")
        out = self.b[self.o:self.o + n]
        self.o += n
        return out

    def unpack(self, fmt: str):
        return struct.unpack("<" + fmt, self.take(struct.calcsize("<" + fmt)))

    def fourcc(self) -> str:
        return self.take(4).decode("latin-1")

    def array(self, dtype, n: int) -> np.ndarray:
        return np.frombuffer(self.take(n * np.dtype(dtype).itemsize), dtype=dtype).copy()

    def vector(self, dtype) -> np.ndarray:
        (n,) = self.unpack("Q")
        return self.array(dtype, n)


def _read_header(r: _Reader) -> Dict:
    d, ntotal, _, _, trained, metric = r.unpack("iqqq?i")
    if metric > 1:
        r.unpack("f")
    return dict(d=d, ntotal=ntotal, is_trained=trained, metric=metric)


def _read_flat(r: _Reader, cc: str) -> Dict:
    h = _read_header(r)
    xb = r.vector(np.float32)
    if xb.size != h["ntotal"] * h["d"]:
        raise FaissFormatError(f"{cc}: {xb.size} floats for ntotal={h['ntotal']} d={h['d']}")
    h["vectors"] = xb.reshape(h["ntotal"], h["d"])
    return h


def read_index(path: str) -> Dict:
    """-> dict(kind="flat"|"ivf", d, ntotal, metric, vectors f32[ntotal,d] in id order, and for ivf: nlist, nprobe,
    centroids f32[nlist,d], assign int32[ntotal], ids int64[ntotal] (the stored ids, = arange for index.add()))."""
    r = _Reader(Path(path).read_bytes())
    cc = r.fourcc()
    if cc in ("IxFI", "IxF2", "IxFl"):
        out = _read_flat(r, cc)
        out["kind"] = "flat"
        return out
    if cc != "IwFl":
        raise FaissFormatError(f"unsupported FAISS index type {cc!r} (IndexFlat / IndexIVFFlat only)")
    h = _read_header(r)
    nlist, nprobe = r.unpack("QQ")
    qcc = r.fourcc()
    if qcc not in ("IxFI", "IxF2", "IxFl"):
        raise FaissFormatError(f"unsupported coarse quantizer {qcc!r} (IndexFlat only)")
    q = _read_flat(r, qcc)
    if q["ntotal"] != nlist or q["d"] != h["d"]:
        raise FaissFormatError("quantizer shape does not match nlist/d")
    (dm_type,) = r.unpack("b")
    r.vector(np.int64)
    if dm_type == 2:
        (n,) = r.unpack("Q")
        r.take(16 * n)
    lcc = r.fourcc()
    d, ntotal = h["d"], h["ntotal"]
    vectors = np.zeros((ntotal, d), dtype=np.float32)
    assign = np.full((ntotal,), -1, dtype=np.int32)
    ids_all = np.arange(ntotal, dtype=np.int64)
    if lcc == "il00":
        sizes = np.zeros(nlist, dtype=np.uint64)
    elif lcc == "ilar":
        nl2, code_size = r.unpack("QQ")
        if nl2 != nlist or code_size != 4 * d:
            raise FaissFormatError(f"inverted lists: nlist={nl2} code_size={code_size} (expected {nlist}, {4 * d})")
        lt = r.fourcc()
        raw = r.vector(np.uint64)
        if lt == "full":
            sizes = raw
        elif lt == "sprs":
            sizes = np.zeros(nlist, dtype=np.uint64)
            sizes[raw[0::2].astype(np.int64)] = raw[1::2]
        else:
            raise FaissFormatError(f"inverted lists: unknown list type {lt!r}")
        if sizes.size != nlist or int(sizes.sum()) != ntotal:
            raise FaissFormatError("inverted list sizes do not add up to ntotal")
        pos = 0
        for c in range(nlist):
            n = int(sizes[c])
            if n == 0:
                continue
            codes = r.array(np.float32, n * d).reshape(n, d)
            ids = r.array(np.int64, n)
            if ids.min() < 0 or ids.max() >= ntotal:
                raise FaissFormatError("stored ids outside [0, ntotal): only add()-style sequential ids are supported")
            vectors[ids] = codes
            assign[ids] = c
            pos += n
    else:
        raise FaissFormatError(f"unsupported inverted-list container {lcc!r}")
    if (assign < 0).any():
        raise FaissFormatError("some ids in [0, ntotal) are in no inverted list")
    h.update(kind="ivf", nlist=int(nlist), nprobe=int(nprobe), centroids=q["vectors"], vectors=vectors, assign=assign,
             ids=ids_all)
    return h


def _header(d: int, ntotal: int, metric: int) -> bytes:
    return struct.pack("<iqqq?i", d, ntotal, 1 << 20, 1 << 20, True, metric)


def _flat_bytes(x: np.ndarray, metric: int) -> bytes:
    x = np.ascontiguousarray(x, dtype=np.float32)
    cc = b"IxFI" if metric == METRIC_INNER_PRODUCT else b"IxF2"
    return cc + _header(x.shape[1], x.shape[0], metric) + struct.pack("<Q", x.size) + x.tobytes()


def write_flat(path: str, vectors: np.ndarray, metric: int = METRIC_INNER_PRODUCT) -> None:
    Path(path).write_bytes(_flat_bytes(vectors, metric))


def write_ivf_flat(path: str, vectors: np.ndarray, centroids: np.ndarray, assign: np.ndarray, nprobe: int,
                   metric: int = METRIC_INNER_PRODUCT) -> None:
    """vectors f32[ntotal,d] in id order (ids = 0..ntotal-1), centroids f32[nlist,d], assign[ntotal] -> IndexIVFFlat"""
    x = np.ascontiguousarray(vectors, dtype=np.float32)
    c = np.ascontiguousarray(centroids, dtype=np.float32)
    a = np.asarray(assign, dtype=np.int64)
    ntotal, d = x.shape
    nlist = c.shape[0]
    parts = [b"IwFl", _header(d, ntotal, metric), struct.pack("<QQ", nlist, int(nprobe)), _flat_bytes(c, metric),
             struct.pack("<b", 0), struct.pack("<Q", 0)]                       # direct map: none, empty array
    order = np.argsort(a, kind="stable")                                        # ids ascending inside a list
    sizes = np.bincount(a, minlength=nlist).astype(np.uint64)
    parts += [b"ilar", struct.pack("<QQ", nlist, 4 * d)]
    if int((sizes > 0).sum()) > nlist // 2:
        parts += [b"full", struct.pack("<Q", nlist), sizes.tobytes()]
    else:
        nz = np.nonzero(sizes)[0]
        pairs = np.stack([nz.astype(np.uint64), sizes[nz]], 1).reshape(-1)
        parts += [b"sprs", struct.pack("<Q", pairs.size), pairs.tobytes()]
    start = 0
    for n in sizes.astype(np.int64):
        if n > 0:
            ids = order[start:start + n].astype(np.int64)
            parts += [x[ids].tobytes(), ids.tobytes()]
            start += n
    Path(path).write_bytes(b"".join(parts))


def sniff(path: str) -> Optional[str]:
    """"rihip" / "faiss" / None by the first bytes of the file"""
    with open(path, "rb") as f:
        head = f.read(8)
    if head == b"RIHIPIDX":
        return "rihip"
    if head[:4] in (b"IwFl", b"IxFI", b"IxF2", b"IxFl"):
        return "faiss"
    return None
