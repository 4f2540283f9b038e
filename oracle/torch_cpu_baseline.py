"""Stock torch-CPU float32 restatement of the reference path, timed beside the GPU numbers (TEST INFRASTRUCTURE:
imported only by bench.py's `cpu_baseline` legs and by tests/).

BASELINE.md §3 / SURVEY.md §8(d) "CPU baseline timed beside it": the same math as the reference, driven by stock torch
CPU ops on the GPU box's host cores, fed from pre-materialised tensors (model-bound) -- and, for the sampled step, the
reference's per-sample Python sampler timed separately (its real, input-bound rate).  faiss / lightgbm are not
installed, so the retrieval baseline is `Q @ X.T` + `topk` in query tiles and the ranker baseline a NumPy tree walk.

Reference lines restated:
  towers            src/models/two_tower.py:19-72        (Embedding -> Linear -> ReLU -> Dropout -> Linear -> normalize)
  bpr_loss          src/models/two_tower.py:117-130
  in_batch_bpr_loss src/models/two_tower.py:132-160      (vectorised closed form: the reference's Python loop over B is
                                                          O(B) kernel launches and would only flatter the GPU)
  step              src/training/train_embeddings.py:183-192 (towers, loss, backward, clip_grad_norm_ 1.0, Adam wd 1e-5)
  sampler           src/training/train_embeddings.py:58-79   (rejection against the user's rated set, per sample)
  search            src/models/faiss_index.py:126-153        (inner product + top-k; brute force here)
  predict           src/models/ranker.py:161-174             (sum of reached leaves)
"""
from __future__ import annotations

import time
from typing import Dict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

N_GENRES = 18


class _Tower(nn.Module):
    def __init__(self, n: int, d: int, hidden: int, extra: int, dropout: float):
        super().__init__()
        self.embedding = nn.Embedding(n + 1, d, padding_idx=0)
        self.mlp = nn.Sequential(nn.Linear(d + extra, hidden), nn.ReLU(), nn.Dropout(dropout), nn.Linear(hidden, d))
        nn.init.xavier_uniform_(self.embedding.weight)

    def forward(self, ids, genres=None):
        x = self.embedding(ids)
        if genres is not None:
            x = torch.cat([x, genres], dim=-1)
        return F.normalize(self.mlp(x), p=2, dim=-1)


class TwoTowerCPU(nn.Module):
    def __init__(self, n_users: int, n_items: int, d: int = 64, hidden: int = 128, dropout: float = 0.1):
        super().__init__()
        self.user_tower = _Tower(n_users, d, hidden, 0, dropout)
        self.item_tower = _Tower(n_items, d, hidden, N_GENRES, dropout)


def _threads() -> int:
    return int(torch.get_num_threads())


def time_sampled_step(B: int = 256, n_users: int = 6040, n_items: int = 3952, d: int = 64, hidden: int = 128,
                      budget_s: float = 5.0, seed: int = 0) -> Dict:
    """BASELINE configs[0]: the reference's training step (sampled negatives, dense Adam + L2), model-bound."""
    torch.manual_seed(seed)
    m = TwoTowerCPU(n_users, n_items, d, hidden)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)
    g = torch.Generator().manual_seed(seed)
    nb = 16
    bs = [(torch.randint(1, n_users + 1, (B,), generator=g), torch.randint(1, n_items + 1, (B,), generator=g),
           (torch.rand((B, N_GENRES), generator=g) < 0.1).float(), torch.randint(1, n_items + 1, (B,), generator=g),
           (torch.rand((B, N_GENRES), generator=g) < 0.1).float()) for _ in range(nb)]

    def step(i):
        u, p, gp, n, gn = bs[i % nb]
        opt.zero_grad()
        U, P, N = m.user_tower(u), m.item_tower(p, gp), m.item_tower(n, gn)
        loss = -F.logsigmoid((U * P).sum(-1) - (U * N).sum(-1)).mean()
        loss.backward()
        nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()

    for i in range(3):
        step(i)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        step(n)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n * B / dt, "unit": "pairs/s", "cores": _threads(), "kind": "port",
            "sample": f"{n} steps of B={B} on {n_users}x{n_items} tables d={d} (stock torch-CPU f32: towers, bpr_loss, "
                      f"autograd, clip_grad_norm_, dense Adam+L2), pre-materialised batches, {dt:.1f}s"}


def time_reference_sampler(n_users: int = 6040, n_items: int = 3952, per_user: int = 165, budget_s: float = 3.0,
                           seed: int = 0) -> Dict:
    """The reference's UserItemDataset.__getitem__ (per-sample Python rejection sampler + tensor construction): the
    rate the reference's DataLoader can feed ONE worker at -- its real bottleneck."""
    rng = np.random.RandomState(seed)
    all_items = np.arange(1, n_items + 1)
    rated = {u: set(rng.choice(all_items, per_user, replace=False).tolist()) for u in range(1, n_users + 1)}
    genre = {int(i): (rng.rand(N_GENRES) < 0.1).astype(np.float32) for i in all_items}
    users = rng.randint(1, n_users + 1, 200_000)
    pos = rng.randint(1, n_items + 1, 200_000)
    zeros = np.zeros(N_GENRES, dtype=np.float32)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < len(users):
        u, p = int(users[n]), int(pos[n])
        r = rated.get(u, set())
        while True:
            neg = int(np.random.choice(all_items))
            if neg not in r:
                break
        _ = (torch.tensor(u, dtype=torch.long), torch.tensor(p, dtype=torch.long),
             torch.tensor(genre.get(p, zeros), dtype=torch.float32), torch.tensor(neg, dtype=torch.long),
             torch.tensor(genre.get(neg, zeros), dtype=torch.float32))
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"{n} samples through the per-sample rejection sampler of one DataLoader worker, {dt:.1f}s"}


def time_inbatch_block(G: int = 65536, blk: int = 2048, d: int = 128, hidden: int = 128, budget_s: float = 8.0,
                       seed: int = 0) -> Dict:
    """A block of `blk` users (and their positives) of one global step against all G in-batch items: towers fwd+bwd
    for the block's rows + the in-batch BPR loss and both gradients in the vectorised closed form, f32.  Same work
    per pair as the GPU step (every pair is scored against all G items)."""
    torch.manual_seed(seed)
    m = TwoTowerCPU(4 * blk, 4 * blk, d, hidden)
    m.train()
    g = torch.Generator().manual_seed(seed)
    I_all = F.normalize(torch.randn((G, d), generator=g), dim=-1)
    scale = 1.0 / (G * (G - 1.0))
    n, t0 = 0, time.perf_counter()
    while True:
        u = torch.randint(1, 4 * blk + 1, (blk,), generator=g)
        p = torch.randint(1, 4 * blk + 1, (blk,), generator=g)
        gp = (torch.rand((blk, N_GENRES), generator=g) < 0.1).float()
        U, P = m.user_tower(u), m.item_tower(p, gp)
        with torch.no_grad():
            I_all[:blk] = P
            S = U @ I_all.T                                        # [blk, G]
            z = S - (U * P).sum(-1, keepdim=True)
            idx = torch.arange(blk)
            loss = (F.softplus(z).sum() - F.softplus(z[idx, idx]).sum()) * scale
            Gm = torch.sigmoid(z).mul_(scale)
            Gm[idx, idx] = 0.0
            r = Gm.sum(1, keepdim=True)
            dU = Gm @ I_all - r * P                                # G_ii = -sum_j G_ij
            dI = Gm.T @ U                                          # this block's partial for all G items
            dI[:blk] -= r * U
        U.backward(dU)
        P.backward(dI[:blk])
        m.zero_grad(set_to_none=True)
        n += blk
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "pairs/s", "cores": _threads(), "kind": "port",
            "sample": f"{n} pairs = {n // blk} blocks of {blk} users x all {G} in-batch items of one step (stock torch-CPU "
                      f"f32: towers fwd+bwd by autograd + vectorised in-batch BPR loss/gradients), {dt:.1f}s; "
                      f"final block loss {float(loss):.4f}"}


def time_retrieval(N: int = 1_000_000, d: int = 128, k: int = 500, tile: int = 256, budget_s: float = 8.0,
                   seed: int = 1) -> Dict:
    g = torch.Generator().manual_seed(seed)
    X = F.normalize(torch.randn((N, d), generator=g), dim=-1)
    n, t0 = 0, None
    for it in range(1000):
        Q = F.normalize(torch.randn((tile, d), generator=g), dim=-1)
        if it == 1:
            n, t0 = 0, time.perf_counter()                        # first tile = warm-up
        s, r = torch.topk(Q @ X.T, k, dim=1)
        n += tile
        if t0 is not None and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "queries/s", "cores": _threads(), "kind": "port",
            "sample": f"{n} queries in tiles of {tile} against {N}x{d} f32 (torch-CPU Q@X.T + topk({k})), {dt:.1f}s"}


def time_tree_walk(forest, n_features: int, n: int = 8192, budget_s: float = 5.0, seed: int = 2) -> Dict:
    from . import gbdt_np as G
    rng = np.random.RandomState(seed)
    X = rng.rand(n, n_features).astype(np.float32)
    done, t0 = 0, time.perf_counter()
    while True:
        G.predict_raw(forest, X)
        done += n
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "candidates/s", "cores": 1, "kind": "port",
            "sample": f"{done} candidates x {len(forest['trees'])} trees (NumPy tree walk, vectorised over candidates), {dt:.1f}s"}
