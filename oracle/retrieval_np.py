"""NumPy restatement of the retrieval stage (TEST INFRASTRUCTURE).

Follows the *wrapper* semantics of /root/reference/src/models/faiss_index.py
(:45-82 build, :88-124 search, :126-153 batch_search).  The arithmetic itself
lives in faiss-cpu (>=1.7.4, requirements.txt:2), which is third-party and not
installed here: **parity unpinned** -- checked through the reference's own
property tests (tests/test_models.py:168-246) and exactness vs brute force.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

F32 = np.float32


def normalize_rows(x: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """faiss_index.py:64-65 / :109-110 / :141-142 (eps 1e-8, not 1e-12)."""
    x = np.asarray(x, dtype=F32)
    n = np.linalg.norm(x, axis=1, keepdims=True)
    return (x / np.maximum(n, eps)).astype(F32)


def topk_ip_exact(Q: np.ndarray, X: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Exact inner-product top-k, descending, ties -> lowest row index.

    Scores are computed in float64 from the float32 inputs and rounded to
    float32 (what an exact-f32 fmaf chain approximates to ~1e-7).
    Returns (scores f32[Q,k], rows i64[Q,k]).
    """
    S = (Q.astype(np.float64) @ X.astype(np.float64).T)
    k = min(k, X.shape[0])
    # stable argsort on -S keeps lowest index first among ties
    order = np.argsort(-S, axis=1, kind="stable")[:, :k]
    sc = np.take_along_axis(S, order, axis=1).astype(F32)
    return sc, order.astype(np.int64)


def topk_ip_exact_f32(Q: np.ndarray, X: np.ndarray, k: int):
    """Same, scores in float32 arithmetic (bit-exact target for integer-valued inputs)."""
    S = (Q.astype(F32) @ X.astype(F32).T).astype(F32)
    k = min(k, X.shape[0])
    order = np.argsort(-S, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(S, order, axis=1), order.astype(np.int64)


# ------------------------------ IVF-Flat (IP) ------------------------------ #
def ivf_assign(X: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """List assignment by max inner product with the centroid (IndexFlatIP quantizer)."""
    return np.argmax(X.astype(np.float64) @ centroids.astype(np.float64).T, axis=1).astype(np.int64)


def ivf_search(Q, X, centroids, assign, nprobe: int, k: int, return_probe: bool = False):
    """IVF-IP search given centroids + assignment: coarse top-nprobe lists by IP (ties -> lowest list id),
    exact scan of those lists, top-k desc (ties -> lowest row); -1 / -inf padding when fewer than k
    (faiss convention kept by faiss_index.py:148-152).  Scores in float64, rounded to float32.
    return_probe: also return (probe lists i64[nq,nprobe], coarse scores f64[nq,nlist])."""
    nq = Q.shape[0]
    k = min(k, X.shape[0])
    assign = np.asarray(assign)
    nlist = centroids.shape[0]
    coarse = Q.astype(np.float64) @ centroids.astype(np.float64).T
    probe = np.argsort(-coarse, axis=1, kind="stable")[:, :nprobe]
    order = np.argsort(assign, kind="stable")                       # rows grouped by list, ascending inside
    starts = np.searchsorted(assign[order], np.arange(nlist + 1))
    out_s = np.full((nq, k), -np.inf, dtype=F32)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        rows = np.concatenate([order[starts[c]:starts[c + 1]] for c in probe[q]])
        if rows.size == 0:
            continue
        s = X[rows].astype(np.float64) @ Q[q].astype(np.float64)
        o = np.lexsort((rows, -s))[:k]
        out_s[q, : o.size] = s[o].astype(F32)
        out_i[q, : o.size] = rows[o]
    if return_probe:
        return out_s, out_i, probe, coarse
    return out_s, out_i


def kmeans_ip(X: np.ndarray, n_lists: int, n_iter: int = 20, seed: int = 1234, init=None) -> np.ndarray:
    """Plain Lloyd k-means used by the build's own IVF trainer (centroids = mean of
    members, assignment by max IP, lowest list id on ties; an empty list keeps its centroid).
    `init` f32[n_lists,d] = starting centroids (default: seeded distinct rows).
    NOT faiss's trainer -- parity unpinned."""
    rng = np.random.RandomState(seed)
    n = X.shape[0]
    if init is not None:
        C = np.asarray(init, dtype=np.float64).copy()
    else:
        C = X[rng.choice(n, n_lists, replace=False)].astype(np.float64).copy()
    for _ in range(n_iter):
        a = np.argmax(X.astype(np.float64) @ C.T, axis=1)
        for c in range(n_lists):
            m = a == c
            if m.any():
                C[c] = X[m].mean(0)
    return C.astype(F32)
