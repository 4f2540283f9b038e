"""Drop-in FAISSIndex backed by the gfx950 inner-product index (no faiss).

Mirrors the reference's src/models/faiss_index.py (:23-228): constructor, build_ivf_index,
search, batch_search, save/load (+ ``<stem>.meta.pkl`` sidecar with the same keys), stats,
set_n_probe, and the ``index`` attribute (``index.ntotal``, writable ``index.nprobe``) that
tests/test_models.py:170-171,:223 read.  Error types/messages follow the reference.

Index files: ``save(path)`` writes the library's own "RIHIPIDX" format by default and a FAISS ``IndexIVFFlat`` /
``IndexFlatIP`` file with ``format="faiss"``; ``load(path)`` sniffs the magic and reads either (faiss_io.py; the FAISS
layout is restated from faiss 1.7.x and unverifiable offline: parity unpinned).  The k-means trainer is
the library's own, so the IVF list membership of an index TRAINED here differs from one trained by faiss; an index
LOADED from a faiss file keeps faiss's centroids and lists.  Limits the reference (faiss) does not have: embed_dim <= 128
(any width: rows are zero-padded to the 32/64/128 kernel width inside the handle), n_lists <= 2048, k <= 16384.
"""
from __future__ import annotations

import ctypes as C
import logging
import pickle
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L

logger = logging.getLogger(__name__)

FAISS_AVAILABLE = True  # name kept for callers that probe it


class _IndexHandle:
    """Owns one rihip ip_index; exposes the two faiss attributes the callers touch."""

    def __init__(self, handle: int, owner: "FAISSIndex"):
        self._h = C.c_void_p(handle)
        self._owner = owner

    @property
    def ntotal(self) -> int:
        return int(L.lib().rihip_ip_index_ntotal(self._h))

    @property
    def nprobe(self) -> int:
        return self._owner.n_probe

    @nprobe.setter
    def nprobe(self, v: int) -> None:
        self._owner.n_probe = int(v)
        L.check(L.lib().rihip_ip_index_set_nprobe(self._h, int(v)), "ip_index_set_nprobe")

    @property
    def is_ivf(self) -> bool:
        return bool(L.lib().rihip_ip_index_is_ivf(self._h))

    def __del__(self):
        try:
            if self._h:
                L.lib().rihip_ip_index_destroy(self._h)
                self._h = None
        except Exception:
            pass


class FAISSIndex:
    def __init__(self, embed_dim: int = 64, n_lists: int = 100, n_probe: int = 10, exact: bool = False):
        """exact=True skips the IVF partition (brute-force inner product; not in the reference)."""
        self.embed_dim = embed_dim
        self.n_lists = n_lists
        self.n_probe = n_probe
        self.exact = exact
        self.index: Optional[_IndexHandle] = None
        self.item_ids: Optional[np.ndarray] = None
        self._item_id_to_faiss_idx: Dict[int, int] = {}
        self._item_ids_dev: Optional[torch.Tensor] = None

    # -- build (faiss_index.py:45-82) ---------------------------------------------------------
    def build_ivf_index(self, embeddings: np.ndarray, item_ids: List[int], kmeans_iters: int = 20,
                        seed: int = 1234) -> None:
        assert embeddings.dtype == np.float32, "Embeddings must be float32"
        assert embeddings.shape[1] == self.embed_dim, (
            f"Expected embed_dim={self.embed_dim}, got {embeddings.shape[1]}"
        )
        norms = np.linalg.norm(embeddings, axis=1, keepdims=True)
        embeddings = np.ascontiguousarray(embeddings / np.maximum(norms, 1e-8), dtype=np.float32)
        x_dev = torch.from_numpy(embeddings).to(L.device())
        self.build_from_device(x_dev, np.array(item_ids, dtype=np.int64), kmeans_iters=kmeans_iters, seed=seed)
        self._item_id_to_faiss_idx = {int(iid): idx for idx, iid in enumerate(item_ids)}

    def build_from_device(self, x_dev: torch.Tensor, item_ids: np.ndarray, kmeans_iters: int = 20,
                          seed: int = 1234, init_centroids: Optional[np.ndarray] = None,
                          centroids: Optional[np.ndarray] = None, assign: Optional[np.ndarray] = None) -> None:
        """Build from already-normalised f32 [N,d] rows on the device (no host round trip).

        init_centroids f32[n_lists,d]: k-means starts from these instead of seeded rows (kmeans_iters=0 partitions
        by them as they are).  centroids + assign int32[N]: inject a trained partition (a FAISS IndexIVFFlat file)."""
        lib = L.lib()
        n = x_dev.shape[0]
        assert x_dev.dtype == torch.float32 and x_dev.shape[1] == self.embed_dim and x_dev.is_contiguous()
        h = C.c_void_p()
        L.check(lib.rihip_ip_index_create(self.embed_dim, C.byref(h)), "ip_index_create")
        self.index = _IndexHandle(h.value, self)
        L.check(lib.rihip_ip_index_set_vectors(self.index._h, x_dev.data_ptr(), n, 1, L.stream_ptr()),
                "ip_index_set_vectors")
        if not self.exact:
            nlist = max(1, min(self.n_lists, n))
            logger.info("Training IVF index on %d vectors (n_lists=%d)...", n, nlist)
            if centroids is not None:
                c = np.ascontiguousarray(centroids, dtype=np.float32)
                a = np.ascontiguousarray(assign, dtype=np.int32)
                assert c.shape == (nlist, self.embed_dim) and a.shape == (n,)
                L.check(lib.rihip_ip_index_set_ivf(self.index._h, nlist, c.ctypes.data, a.ctypes.data, L.stream_ptr()),
                        "ip_index_set_ivf")
            elif init_centroids is not None:
                c = np.ascontiguousarray(init_centroids, dtype=np.float32)
                assert c.shape == (nlist, self.embed_dim)
                L.check(lib.rihip_ip_index_train_ivf_from(self.index._h, nlist, kmeans_iters, c.ctypes.data,
                                                          L.stream_ptr()), "ip_index_train_ivf_from")
            else:
                L.check(lib.rihip_ip_index_train_ivf(self.index._h, nlist, kmeans_iters, seed, L.stream_ptr()),
                        "ip_index_train_ivf")
        L.check(lib.rihip_ip_index_set_nprobe(self.index._h, int(self.n_probe)), "ip_index_set_nprobe")
        self.item_ids = np.asarray(item_ids, dtype=np.int64)
        self._item_ids_dev = torch.from_numpy(self.item_ids).to(x_dev.device)
        logger.info("Index built: %d vectors, %d lists, probe=%d", self.index.ntotal, self.n_lists, self.n_probe)

    # -- trained state (what faiss exposes as index.quantizer / index.invlists) ------------------
    def centroids(self) -> np.ndarray:
        """f32 [nlist,d] coarse centroids of the IVF partition."""
        nl = int(L.lib().rihip_ip_index_nlist(self.index._h))
        c = np.empty((nl, self.embed_dim), dtype=np.float32)
        L.check(L.lib().rihip_ip_index_get_ivf(self.index._h, c.ctypes.data, None), "ip_index_get_ivf")
        return c

    def list_assignment(self) -> np.ndarray:
        """int32 [N]: the inverted list every stored row (insertion order) belongs to."""
        a = np.empty(self.index.ntotal, dtype=np.int32)
        L.check(L.lib().rihip_ip_index_get_ivf(self.index._h, None, a.ctypes.data), "ip_index_get_ivf")
        return a

    def reconstruct(self) -> np.ndarray:
        """f32 [N,d]: the stored (normalised) vectors in insertion order (faiss reconstruct_n(0, ntotal))."""
        x = np.empty((self.index.ntotal, self.embed_dim), dtype=np.float32)
        L.check(L.lib().rihip_ip_index_reconstruct(self.index._h, x.ctypes.data), "ip_index_reconstruct")
        return x

    def assign_lists(self, x_dev: torch.Tensor) -> torch.Tensor:
        """int32 [n] on device: arg-max-IP list of each f32 [n,d] device row (the IndexFlatIP quantizer)."""
        x = x_dev.to(dtype=torch.float32).contiguous()
        out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
        L.check(L.lib().rihip_ip_index_assign(self.index._h, x.data_ptr(), x.shape[0], out.data_ptr(), L.stream_ptr()),
                "ip_index_assign")
        return out

    # -- search (faiss_index.py:88-153) -------------------------------------------------------
    def _search_device(self, q_dev: torch.Tensor, k: int, item_ids: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """q_dev: normalised f32 [nq,d] on device -> (scores [nq,k], rows [nq,k]) on device; item_ids=True: the rows
        come back as item ids (faiss_index.py:123,148-152), mapped inside the search's last kernel."""
        lib = L.lib()
        nq = q_dev.shape[0]
        L.check(lib.rihip_ip_index_set_id_map(self.index._h, self._item_ids_dev.data_ptr() if item_ids else None),
                "ip_index_set_id_map")
        if k > int(lib.rihip_ip_index_max_k()):
            raise ValueError(f"k={k} exceeds the device select buffer ({int(lib.rihip_ip_index_max_k())}); "
                             "the reference (faiss) has no such limit -- see INTEGRATION.md")
        scores = torch.empty((nq, k), dtype=torch.float32, device=q_dev.device)
        rows = torch.empty((nq, k), dtype=torch.int64, device=q_dev.device)
        L.check(lib.rihip_ip_index_search(self.index._h, q_dev.data_ptr(), nq, k, scores.data_ptr(), rows.data_ptr(),
                                          L.stream_ptr()), "ip_index_search")
        return scores, rows

    def set_deferred_check(self, enable: bool) -> None:
        """Serving chains (not in the reference): a thresholded IVF search then returns without its host
        synchronisation; call `finish_search()` after enqueueing the consumers of the result."""
        L.check(L.lib().rihip_ip_index_set_deferred_check(self.index._h, 1 if enable else 0), "ip_index_set_deferred_check")

    def finish_search(self) -> int:
        """-> number of queries of the last deferred search that had to be re-done exactly (their output rows were
        rewritten AFTER anything enqueued behind the search ran: run those consumers again); 0 almost always."""
        n = C.c_int(0)
        L.check(L.lib().rihip_ip_index_search_finish(self.index._h, C.byref(n), L.stream_ptr()), "ip_index_search_finish")
        return int(n.value)

    def search_pending(self) -> bool:
        return bool(L.lib().rihip_ip_index_search_pending(self.index._h))

    def last_fail_count(self) -> int:
        """after the REPLAY of a captured chain that holds a deferred search: synchronises, -> its failure count"""
        n = C.c_int(0)
        L.check(L.lib().rihip_ip_index_last_fail_count(self.index._h, C.byref(n), L.stream_ptr()), "ip_index_last_fail_count")
        return int(n.value)

    def search(self, query_vector: np.ndarray, k: int = 500) -> Tuple[np.ndarray, np.ndarray]:
        if self.index is None:
            raise RuntimeError("Index not built. Call build_ivf_index() first.")
        query = np.atleast_2d(query_vector).astype(np.float32)
        norm = np.linalg.norm(query, axis=1, keepdims=True)
        query = query / np.maximum(norm, 1e-8)
        k = min(k, self.index.ntotal)
        q_dev = torch.from_numpy(np.ascontiguousarray(query[:1])).to(L.device())
        scores, rows = self._search_device(q_dev, k)
        distances = scores[0].cpu().numpy()
        faiss_indices = rows[0].cpu().numpy()
        valid_mask = faiss_indices >= 0
        distances = distances[valid_mask]
        faiss_indices = faiss_indices[valid_mask]
        return distances, self.item_ids[faiss_indices]

    def batch_search(self, query_vectors: np.ndarray, k: int = 500) -> Tuple[np.ndarray, np.ndarray]:
        if self.index is None:
            raise RuntimeError("Index not built.")
        queries = query_vectors.astype(np.float32)
        norms = np.linalg.norm(queries, axis=1, keepdims=True)
        queries = queries / np.maximum(norms, 1e-8)
        k = min(k, self.index.ntotal)
        q_dev = torch.from_numpy(np.ascontiguousarray(queries)).to(L.device())
        scores, rows = self._search_device(q_dev, k, item_ids=True)
        return scores.cpu().numpy(), rows.cpu().numpy()

    def batch_search_device(self, queries: torch.Tensor, k: int = 500, normalized: bool = False
                            ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Device-resident batch_search (not in the reference): queries f32 [nq,d] on the HIP device;
        returns (scores, item_ids) on device, -1 padded.  This is the QPS benchmark entry."""
        if self.index is None:
            raise RuntimeError("Index not built.")
        q = queries.to(dtype=torch.float32).contiguous()
        if not normalized:
            q = q / torch.clamp(torch.linalg.norm(q, dim=1, keepdim=True), min=1e-8)
        k = min(k, self.index.ntotal)
        return self._search_device(q, k, item_ids=True)

    # -- persistence (faiss_index.py:159-205) -------------------------------------------------
    def save(self, path: str, format: str = "rihip") -> None:
        """format="faiss": a file faiss.read_index can open (IndexIVFFlat, or IndexFlatIP for exact=True)."""
        save_path = Path(path)
        save_path.parent.mkdir(parents=True, exist_ok=True)
        if format == "faiss":
            from . import faiss_io
            if self.index.is_ivf:
                faiss_io.write_ivf_flat(str(save_path), self.reconstruct(), self.centroids(), self.list_assignment(),
                                        self.n_probe)
            else:
                faiss_io.write_flat(str(save_path), self.reconstruct())
        elif format == "rihip":
            L.check(L.lib().rihip_ip_index_save(self.index._h, str(save_path).encode()), "ip_index_save")
        else:
            raise ValueError(f"unknown index file format {format!r}")
        meta_path = save_path.with_suffix(".meta.pkl")
        with open(meta_path, "wb") as f:
            pickle.dump(
                {
                    "item_ids": self.item_ids,
                    "item_id_to_faiss_idx": self._item_id_to_faiss_idx,
                    "embed_dim": self.embed_dim,
                    "n_lists": self.n_lists,
                    "n_probe": self.n_probe,
                },
                f,
            )
        logger.info("Saved index to %s (meta: %s)", save_path, meta_path)

    @classmethod
    def load(cls, path: str) -> "FAISSIndex":
        load_path = Path(path)
        if not load_path.exists():
            raise FileNotFoundError(f"FAISS index not found at {load_path}")
        meta_path = load_path.with_suffix(".meta.pkl")
        with open(meta_path, "rb") as f:
            meta = pickle.load(f)  # sidecar written by save() above
        obj = cls(embed_dim=meta["embed_dim"], n_lists=meta["n_lists"], n_probe=meta["n_probe"])
        from . import faiss_io
        if faiss_io.sniff(str(load_path)) == "faiss":        # written by faiss.write_index (or save(format="faiss"))
            f = faiss_io.read_index(str(load_path))
            if f["metric"] != faiss_io.METRIC_INNER_PRODUCT:
                raise ValueError("only METRIC_INNER_PRODUCT indexes are supported (faiss_index.py:70-72)")
            assert f["d"] == obj.embed_dim, (f["d"], obj.embed_dim)
            obj.exact = f["kind"] == "flat"
            x_dev = torch.from_numpy(f["vectors"]).to(L.device())
            ids = np.asarray(meta["item_ids"], dtype=np.int64)
            if obj.exact:
                obj.build_from_device(x_dev, ids)
            else:
                nlist, n = int(f["nlist"]), int(f["ntotal"])
                max_lists = 2048      # csrc/ip_index.h NLIST_MAX (probe bitset of the list-major scan)
                if nlist > max_lists:
                    raise faiss_io.FaissFormatError(
                        f"{load_path}: IndexIVFFlat with nlist={nlist} > {max_lists} lists is not supported by the HIP index")
                if nlist > n:
                    # more lists than vectors (faiss allows it; every search then probes mostly empty lists): serve the
                    # file's vectors exactly instead of failing
                    logger.warning("%s: nlist=%d > ntotal=%d, loading as a flat (exact) index", load_path, nlist, n)
                    obj.exact = True
                    obj.build_from_device(x_dev, ids)
                else:
                    obj.n_lists = nlist
                    obj.build_from_device(x_dev, ids, centroids=f["centroids"], assign=f["assign"])
                    # the sidecar's n_probe is what the reference restores (faiss_index.py:199-201); a file written by
                    # faiss itself carries its own nprobe, used only when the sidecar has none
                    if meta.get("n_probe") is None and f.get("nprobe"):
                        obj.set_n_probe(int(f["nprobe"]))
            obj._item_id_to_faiss_idx = meta["item_id_to_faiss_idx"]
            return obj
        h = C.c_void_p()
        L.check(L.lib().rihip_ip_index_load(str(load_path).encode(), C.byref(h)), "ip_index_load")
        obj.index = _IndexHandle(h.value, obj)
        obj.index.nprobe = meta["n_probe"]
        obj.item_ids = np.asarray(meta["item_ids"], dtype=np.int64)
        obj._item_ids_dev = torch.from_numpy(obj.item_ids).to(L.device())
        obj._item_id_to_faiss_idx = meta["item_id_to_faiss_idx"]
        obj.exact = not obj.index.is_ivf
        return obj

    # -- utilities (faiss_index.py:211-228) ---------------------------------------------------
    def stats(self) -> Dict:
        if self.index is None:
            return {"status": "not built"}
        return {
            "n_vectors": int(self.index.ntotal),
            "embed_dim": self.embed_dim,
            "n_lists": self.n_lists,
            "n_probe": self.n_probe,
            "metric": "inner_product",
            "n_item_ids": len(self.item_ids) if self.item_ids is not None else 0,
        }

    def set_n_probe(self, n_probe: int) -> None:
        self.n_probe = n_probe
        if self.index is not None:
            self.index.nprobe = n_probe
