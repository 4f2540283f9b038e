"""NumPy restatement of the Two-Tower BPR training path (TEST INFRASTRUCTURE).

Follows /root/reference:
  src/models/two_tower.py:39-42    UserTower.forward
  src/models/two_tower.py:68-72    ItemTower.forward
  src/models/two_tower.py:117-130  bpr_loss
  src/models/two_tower.py:132-160  in_batch_bpr_loss (closed form of the loop)
  src/training/train_embeddings.py:160-161,189-197  Adam(+L2) / clip / cosine

Pinned against the imported reference by oracle/make_golden.py ->
tests/golden/two_tower_*.npz (checked in tests/test_oracle.py).

All arrays float32 unless noted; ids int64.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

F32 = np.float32
NORM_EPS = 1e-12  # F.normalize default eps (two_tower.py:42,72)


# --------------------------------------------------------------------------- #
# Dropout mask: counter-based, shared bit-for-bit with the HIP kernels         #
# (recommendit_amd/csrc/common.h: rihip_keep()).  The reference uses torch's   #
# CPU Philox stream, which no other implementation can reproduce bit-for-bit;  #
# parity with the reference is therefore stated at dropout=0 / eval, and the   #
# train-mode path is checked against THIS mask definition.                     #
# --------------------------------------------------------------------------- #
def splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x = x * np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x = x * np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x = x * np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def dropout_keep_mask(seed: int, row0: int, n_rows: int, n_cols: int, p: float) -> np.ndarray:
    """keep[r, c] for global element index (row0 + r) * n_cols + c  (recommendit_amd/csrc/common.h: rihip_keep)."""
    if p <= 0.0:
        return np.ones((n_rows, n_cols), dtype=bool)
    idx = (np.arange(row0, row0 + n_rows, dtype=np.uint64)[:, None] * np.uint64(n_cols)
           + np.arange(n_cols, dtype=np.uint64)[None, :])
    sm = int(splitmix64(np.array([seed], dtype=np.uint64))[0])
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32) ^ np.uint32(sm & 0xFFFFFFFF)
    hi = (idx >> np.uint64(32)).astype(np.uint32) ^ np.uint32(sm >> 32)
    with np.errstate(over="ignore"):
        h = lowbias32(lo ^ lowbias32(hi + np.uint32(0x9E3779B9)))
    thresh = np.uint32(min(int(p * 16777216.0), 16777215))
    return (h >> np.uint32(8)) >= thresh


# --------------------------------------------------------------------------- #
# Towers                                                                       #
# --------------------------------------------------------------------------- #
@dataclass
class TowerParams:
    table: np.ndarray  # [n+1, d]
    W1: np.ndarray     # [H, K1]   K1 = d (user) or d+18 (item)
    b1: np.ndarray     # [H]
    W2: np.ndarray     # [d, H]
    b2: np.ndarray     # [d]


def tower_forward(
    p: TowerParams,
    ids: np.ndarray,
    genres: Optional[np.ndarray] = None,
    keep_mask: Optional[np.ndarray] = None,
    dropout_p: float = 0.0,
) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    """two_tower.py:39-42 / :68-72.  keep_mask given => training-mode dropout."""
    x = p.table[ids]                                   # embedding gather
    if genres is not None:
        x = np.concatenate([x, genres.astype(F32)], axis=-1)   # torch.cat (:70)
    pre = x @ p.W1.T + p.b1                            # Linear 1
    h = np.maximum(pre, F32(0))                        # ReLU
    if keep_mask is not None and dropout_p > 0.0:      # Dropout between ReLU and Linear 2
        scale = F32(1.0 / (1.0 - dropout_p))
        h = np.where(keep_mask, h * scale, F32(0)).astype(F32)
    y = h @ p.W2.T + p.b2                              # Linear 2
    nrm = np.sqrt((y.astype(F32) ** 2).sum(-1, keepdims=True, dtype=F32))
    denom = np.maximum(nrm, F32(NORM_EPS))
    out = (y / denom).astype(F32)
    cache = dict(x=x.astype(F32), h=h.astype(F32), y=y.astype(F32), denom=denom.astype(F32), out=out,
                 keep=keep_mask, dropout_p=dropout_p)
    return out, cache


def tower_backward(p: TowerParams, cache: Dict[str, np.ndarray], gout: np.ndarray):
    """Hand-derived backward of tower_forward (autograd of two_tower.py:39-42).

    Returns dX_emb [B,d] (per-sample embedding-row grads, to be scatter-added),
    dW1, db1, dW2, db2.
    """
    out, denom, h, x = cache["out"], cache["denom"], cache["h"], cache["x"]
    d = p.table.shape[1]
    # normalise backward: y/max(|y|,eps); for |y|>eps: (g - out*(out.g))/|y|
    dot = (gout * out).sum(-1, keepdims=True, dtype=F32)
    clamped = (denom <= F32(NORM_EPS))
    gy = np.where(clamped, gout / denom, (gout - out * dot) / denom).astype(F32)
    dW2 = gy.T @ h
    db2 = gy.sum(0, dtype=F32)
    dh = gy @ p.W2
    # h already carries relu*dropout*scale: h>0 <=> relu active and kept
    scale = F32(1.0 / (1.0 - cache["dropout_p"])) if (cache["keep"] is not None and cache["dropout_p"] > 0) else F32(1)
    dpre = np.where(h > 0, dh * scale, F32(0)).astype(F32)
    dW1 = dpre.T @ x
    db1 = dpre.sum(0, dtype=F32)
    dx = dpre @ p.W1
    return dx[:, :d].astype(F32), dW1.astype(F32), db1.astype(F32), dW2.astype(F32), db2.astype(F32)


def embedding_scatter_add(n_rows: int, ids: np.ndarray, dX: np.ndarray) -> np.ndarray:
    """Dense embedding grad (nn.Embedding backward, padding_idx=0 row forced to 0)."""
    g = np.zeros((n_rows, dX.shape[1]), dtype=F32)
    np.add.at(g, ids, dX)
    g[0] = 0
    return g


# --------------------------------------------------------------------------- #
# Losses                                                                       #
# --------------------------------------------------------------------------- #
def softplus(z):
    z = np.asarray(z)
    return np.maximum(z, 0) + np.log1p(np.exp(-np.abs(z)))


def sigmoid(z):
    z = np.asarray(z, dtype=np.float64)
    return 1.0 / (1.0 + np.exp(-z))


def bpr_loss(U, P, N):
    """two_tower.py:127-129: mean softplus(-(u.p - u.n)) and grads."""
    B = U.shape[0]
    delta = (U * P).sum(-1, dtype=F32) - (U * N).sum(-1, dtype=F32)
    loss = softplus(-delta.astype(np.float64)).mean()
    w = (-sigmoid(-delta) / B).astype(F32)[:, None]       # dL/d delta
    dU = w * (P - N)
    dP = w * U
    dN = -w * U
    return F32(loss), dU.astype(F32), dP.astype(F32), dN.astype(F32)


def in_batch_bpr_loss(U, I, owner_offset: int = 0, n_global: Optional[int] = None):
    """Closed form of two_tower.py:132-160.

    Square case (owner_offset=0, I.shape[0]==U.shape[0]):
        L = (1/B) sum_i (1/(B-1)) sum_{j!=i} softplus(s_ij - s_ii)
    Rectangular case (multi-GPU): U are local users [Bl,d] whose positives are
    I[owner_offset + i]; B := n_global (= I.shape[0]).  Returned loss is the
    local partial sum already divided by B(B-1); dI is the partial gradient from
    the local users.
    """
    Bl = U.shape[0]
    B = I.shape[0] if n_global is None else n_global
    S = (U.astype(np.float64) @ I.astype(np.float64).T)
    diag_idx = owner_offset + np.arange(Bl)
    pos = S[np.arange(Bl), diag_idx][:, None]
    Z = S - pos
    mask = np.ones_like(S, dtype=bool)
    mask[np.arange(Bl), diag_idx] = False
    c = 1.0 / (B * (B - 1))
    loss = (softplus(Z) * mask).sum() * c
    G = sigmoid(Z) * mask * c
    r = G.sum(1)
    G[np.arange(Bl), diag_idx] = -r
    dU = G @ I.astype(np.float64)
    dI = G.T @ U.astype(np.float64)
    return F32(loss), dU.astype(F32), dI.astype(F32)


def in_batch_bpr_loss_loop(U, I):
    """Literal restatement of the reference's python loop (small B only)."""
    B = U.shape[0]
    S = U.astype(np.float64) @ I.astype(np.float64).T
    tot = 0.0
    for i in range(B):
        m = np.ones(B, dtype=bool)
        m[i] = False
        margins = S[i, i] - S[i][m]
        tot += softplus(-margins).mean()
    return F32(tot / B)


# --------------------------------------------------------------------------- #
# Optimiser: clip_grad_norm_(1.0) + Adam(weight_decay) + CosineAnnealingLR     #
# (train_embeddings.py:160-161, :191-192, :197)                                #
# --------------------------------------------------------------------------- #
def clip_coef(grads, max_norm=1.0):
    tot = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    return min(1.0, max_norm / (tot + 1e-6)), tot


def adam_step(p, g, m, v, step, lr, wd=1e-5, b1=0.9, b2=0.999, eps=1e-8, clip=1.0):
    """torch.optim.Adam single-tensor semantics (coupled L2, not AdamW)."""
    g = (g * F32(clip)).astype(F32)
    if wd != 0.0:
        g = g + F32(wd) * p
    m[:] = F32(b1) * m + F32(1 - b1) * g
    v[:] = F32(b2) * v + F32(1 - b2) * g * g
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    step_size = lr / bc1
    denom = np.sqrt(v) / F32(math.sqrt(bc2)) + F32(eps)
    p[:] = p - F32(step_size) * (m / denom)
    return p, m, v


def cosine_lr(lr0: float, epoch: int, t_max: int) -> float:
    """CosineAnnealingLR closed form, eta_min=0; epoch = number of scheduler.step() calls."""
    return 0.5 * lr0 * (1.0 + math.cos(math.pi * epoch / t_max))


def adam_rows_sparse(p, m, v, ids_unique, g_rows, step, lr, wd=1e-5, b1=0.9, b2=0.999, eps=1e-8, clip=1.0):
    """Row-wise sparse Adam: dense-Adam arithmetic applied to touched rows only.

    Deviation from the reference (SURVEY.md §7 hard part 1): untouched rows do
    not decay / move.  `step` is the global step (one bias-correction clock).
    """
    for r, g in zip(ids_unique, g_rows):
        adam_step(p[r], g.copy(), m[r], v[r], step, lr, wd, b1, b2, eps, clip)
    return p, m, v
