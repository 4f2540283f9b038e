"""Drop-in Two-Tower model whose forward/backward run in the gfx950 HIP library.

Mirrors the class surface of the reference's src/models/two_tower.py (UserTower :19-42,
ItemTower :45-72, TwoTowerModel :75-251): same constructor arguments, attribute names,
``state_dict`` keys (``{user,item}_tower.embedding.weight``, ``.mlp.0.*``, ``.mlp.3.*``),
checkpoint dict keys, and method signatures -- so EmbeddingTrainer
(src/training/train_embeddings.py:183-192), IndexBuilder (src/training/build_index.py:86-105),
RecommendationPipeline (src/serving/recommender.py) and tests/test_models.py run unchanged.

Parameters live on the HIP device.  Inputs given on the CPU are moved to the device and
results returned on the caller's device.  There is no CPU compute path: without a HIP device
every forward raises RuntimeError.
"""
from __future__ import annotations

import logging
import os
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L

logger = logging.getLogger(__name__)

N_GENRES = 18  # two_tower.py:16

_CHECK_IDS = os.environ.get("RECOMMENDIT_CHECK_IDS", "0") == "1"  # device-side id range check (forces a sync)


def _next_seed() -> int:
    """Per-call dropout seed drawn from torch's global generator (so torch.manual_seed governs it)."""
    return int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())


class _TowerFn(torch.autograd.Function):
    """out = normalize(Linear2(dropout(relu(Linear1(cat(E[ids], genres))))))  -- one HIP kernel each way."""

    @staticmethod
    def forward(ctx, table, W1, b1, W2, b2, ids, genres, training, p, seed):
        lib = L.lib()
        dev = L.device()
        for t in (table, W1, b1, W2, b2):
            if t.device.type != "cuda":
                raise RuntimeError("recommendit_amd: tower parameters must live on the HIP device (model.to('cuda'))")
        table_c, W1c, b1c, W2c, b2c = (t.detach().contiguous() for t in (table, W1, b1, W2, b2))
        if ids.device.type == "cpu" and ids.numel() > 0:  # nn.Embedding raises IndexError on CPU ids
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= table.shape[0]:
                raise IndexError(f"index out of range in embedding lookup (table has {table.shape[0]} rows)")
        ids_d = L.i64c(ids)
        g_d = L.f32c(genres) if genres is not None else None
        B = ids_d.numel()
        d = table_c.shape[1]
        H = W1c.shape[0]
        if g_d is not None and tuple(g_d.shape) != (B, N_GENRES):
            raise ValueError(f"genre_vectors must be [{B}, {N_GENRES}], got {tuple(g_d.shape)}")
        if not lib.rihip_tower_shape_ok(d, H):
            raise RuntimeError(f"recommendit_amd: (embed_dim={d}, hidden_dim={H}) unsupported: both must be multiples of "
                               "16 up to 256")
        out = torch.empty((B, d), dtype=torch.float32, device=dev)
        hid = torch.empty((B, H), dtype=torch.float32, device=dev)
        denom = torch.empty((B,), dtype=torch.float32, device=dev)
        err = torch.zeros((1,), dtype=torch.int32, device=dev)
        ws = torch.empty((lib.rihip_tower_forward_workspace_floats(d, H, 0 if g_d is None else 1),),
                         dtype=torch.float32, device=dev)
        L.check(lib.rihip_tower_forward(table_c.data_ptr(), table_c.shape[0], ids_d.data_ptr(), L.ptr(g_d), B, d, H,
                                        W1c.data_ptr(), b1c.data_ptr(), W2c.data_ptr(), b2c.data_ptr(),
                                        1 if training else 0, float(p), seed, 0, out.data_ptr(), hid.data_ptr(),
                                        denom.data_ptr(), err.data_ptr(), ws.data_ptr(), None, L.stream_ptr()),
                "tower_forward")
        ctx.save_for_backward(table_c, W1c, W2c, ids_d, g_d, out, hid, denom)
        ctx.scale = 1.0 / (1.0 - p) if (training and p > 0) else 1.0
        if _CHECK_IDS and int(err.item()) != 0:
            raise IndexError(f"index out of range in embedding lookup (table has {table_c.shape[0]} rows)")
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.lib()
        table, W1, W2, ids, genres, out, hid, denom = ctx.saved_tensors
        dev = table.device
        B, d = out.shape
        H = W1.shape[0]
        K1 = W1.shape[1]
        if B == 0:  # empty batch: all gradients are zero
            z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)
            return torch.zeros_like(table), z(H, K1), z(H), z(d, H), z(d), None, None, None, None, None
        gout = gout.to(device=dev, dtype=torch.float32).contiguous()
        dX = torch.empty((B, d), dtype=torch.float32, device=dev)
        dW1 = torch.empty((H, K1), dtype=torch.float32, device=dev)
        db1 = torch.empty((H,), dtype=torch.float32, device=dev)
        dW2 = torch.empty((d, H), dtype=torch.float32, device=dev)
        db2 = torch.empty((d,), dtype=torch.float32, device=dev)
        nws = lib.rihip_tower_backward_workspace_floats(B, d, H, 1 if genres is not None else 0)
        ws = torch.empty((nws,), dtype=torch.float32, device=dev)
        st = L.stream_ptr()
        L.check(lib.rihip_tower_backward(table.data_ptr(), table.shape[0], ids.data_ptr(), L.ptr(genres), B, d, H,
                                         W1.data_ptr(), W2.data_ptr(), gout.data_ptr(), out.data_ptr(),
                                         denom.data_ptr(), hid.data_ptr(), float(ctx.scale), dX.data_ptr(),
                                         dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), 0,
                                         ws.data_ptr(), st), "tower_backward")
        # dense embedding gradient (stock torch.optim.Adam / clip_grad_norm_ expect it; train_embeddings.py:160,191)
        dtable = torch.zeros_like(table)
        L.check(lib.rihip_embedding_scatter_add(dtable.data_ptr(), table.shape[0], ids.data_ptr(), dX.data_ptr(), B, d,
                                                st), "embedding_scatter_add")
        return dtable, dW1, db1, dW2, db2, None, None, None, None, None


def _run_tower(mod: "nn.Module", ids: torch.Tensor, genres: Optional[torch.Tensor]) -> torch.Tensor:
    in_dev = ids.device
    lin1, lin2 = mod.mlp[0], mod.mlp[3]
    p = float(mod.mlp[2].p)
    training = bool(mod.training and p > 0.0)
    seed = _next_seed() if training else 0
    shape = tuple(ids.shape)
    out = _TowerFn.apply(mod.embedding.weight, lin1.weight, lin1.bias, lin2.weight, lin2.bias, ids.reshape(-1),
                         None if genres is None else genres.reshape(-1, N_GENRES), training, p, seed)
    out = out.reshape(*shape, out.shape[-1])
    return out if in_dev.type == "cuda" else out.to(in_dev)


class UserTower(nn.Module):
    """two_tower.py:19-42 -- same sub-module names, so state_dict keys match."""

    def __init__(self, n_users: int, embed_dim: int, hidden_dim: int = 128, dropout: float = 0.1, device=None):
        super().__init__()
        self.embedding = nn.Embedding(n_users + 1, embed_dim, padding_idx=0, device=device)
        self.mlp = nn.Sequential(
            nn.Linear(embed_dim, hidden_dim, device=device),
            nn.ReLU(),
            nn.Dropout(dropout),
            nn.Linear(hidden_dim, embed_dim, device=device),
        )
        nn.init.xavier_uniform_(self.embedding.weight)  # overwrites the padding zero-row, like the reference (:36-37)

    def forward(self, user_ids: torch.Tensor) -> torch.Tensor:
        return _run_tower(self, user_ids, None)


class ItemTower(nn.Module):
    """two_tower.py:45-72."""

    def __init__(self, n_items: int, embed_dim: int, hidden_dim: int = 128, dropout: float = 0.1, device=None):
        super().__init__()
        self.embedding = nn.Embedding(n_items + 1, embed_dim, padding_idx=0, device=device)
        self.mlp = nn.Sequential(
            nn.Linear(embed_dim + N_GENRES, hidden_dim, device=device),
            nn.ReLU(),
            nn.Dropout(dropout),
            nn.Linear(hidden_dim, embed_dim, device=device),
        )
        nn.init.xavier_uniform_(self.embedding.weight)

    def forward(self, item_ids: torch.Tensor, genre_vectors: torch.Tensor) -> torch.Tensor:
        return _run_tower(self, item_ids, genre_vectors)


class _BprPairFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, P, N):
        lib = L.lib()
        dev = L.device()
        Uc, Pc, Nc = L.f32c(U), L.f32c(P), L.f32c(N)
        B, d = Uc.shape
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dU, dP, dN = torch.empty_like(Uc), torch.empty_like(Pc), torch.empty_like(Nc)
        ws = torch.empty((1024,), dtype=torch.float64, device=dev)
        L.check(lib.rihip_bpr_pair_loss(Uc.data_ptr(), Pc.data_ptr(), Nc.data_ptr(), B, d, loss.data_ptr(),
                                        dU.data_ptr(), dP.data_ptr(), dN.data_ptr(), ws.data_ptr(), L.stream_ptr()),
                "bpr_pair_loss")
        ctx.save_for_backward(dU, dP, dN)
        ctx.in_devs = (U.device, P.device, N.device)
        return loss

    @staticmethod
    def backward(ctx, g):
        dU, dP, dN = ctx.saved_tensors
        g = g.to(dU.device)
        return tuple((t * g).to(dv) for t, dv in zip((dU, dP, dN), ctx.in_devs))


def inbatch_loss_and_grads(U: torch.Tensor, I: torch.Tensor, precision: int = 0, owner_offset: int = 0,
                           n_global: Optional[int] = None, pos: Optional[torch.Tensor] = None,
                           r: Optional[torch.Tensor] = None, users_global: Optional[torch.Tensor] = None,
                           item_owner_offset: Optional[int] = None, store_g: Optional[bool] = None):
    """Single-GPU closed-form in-batch BPR: returns (loss, dU, dI) for square U,I [B,d].
    store_g (exact f32 only; default: on while the B x B matrix stays under 4 GiB): the user pass keeps G in HBM and
    dI is a plain G^T.U product instead of a second score sweep."""
    lib = L.lib()
    dev = L.device()
    Uc, Ic = L.f32c(U), L.f32c(I)
    B, d = Uc.shape
    st = L.stream_ptr()
    posv = torch.empty((B,), dtype=torch.float32, device=dev)
    L.check(lib.rihip_rowdot(Uc.data_ptr(), Ic.data_ptr(), B, 0, d, posv.data_ptr(), st), "rowdot")
    dU, dI = torch.empty_like(Uc), torch.empty_like(Ic)
    rv = torch.empty((B,), dtype=torch.float32, device=dev)
    part = torch.zeros((lib.rihip_inbatch_workspace_doubles(B),), dtype=torch.float64, device=dev)
    ws = torch.empty((lib.rihip_inbatch_workspace_floats(B, B, d),), dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    ng = lib.rihip_inbatch_gmat_floats(B, B)
    if store_g is None:
        store_g = 4 * ng <= (4 << 30)
    if d not in (32, 64, 128):      # widths without a tuned instantiation: the runtime-width two-sweep kernel
        store_g = False
    if store_g and precision in (0, 2):
        gm = torch.empty((ng,), dtype=torch.float32, device=dev)
        L.check(lib.rihip_inbatch_user_pass(Uc.data_ptr(), B, 0, Ic.data_ptr(), B, 0, d, posv.data_ptr(), B,
                                            dU.data_ptr(), rv.data_ptr(), part.data_ptr(), ws.data_ptr(),
                                            gm.data_ptr(), precision, st), "inbatch_user_pass")
        L.check(lib.rihip_inbatch_item_pass(gm.data_ptr(), Uc.data_ptr(), B, 0, B, 0, d, rv.data_ptr(), B, dI.data_ptr(),
                                            ws.data_ptr(), precision, st), "inbatch_item_pass")
        L.check(lib.rihip_sum_partials(part.data_ptr(), lib.rihip_inbatch_loss_parts(B, B), 1.0 / (B * (B - 1.0)),
                                       loss.data_ptr(), st), "sum_partials")
        return loss, dU, dI
    L.check(lib.rihip_inbatch_sweep(1, Uc.data_ptr(), B, 0, Ic.data_ptr(), B, 0, d, posv.data_ptr(), None, B,
                                    dU.data_ptr(), rv.data_ptr(), part.data_ptr(), ws.data_ptr(), precision, st),
            "inbatch_sweep(user)")
    L.check(lib.rihip_inbatch_sweep(0, Ic.data_ptr(), B, 0, Uc.data_ptr(), B, 0, d, posv.data_ptr(), rv.data_ptr(), B,
                                    dI.data_ptr(), None, None, ws.data_ptr(), precision, st), "inbatch_sweep(item)")
    L.check(lib.rihip_sum_partials(part.data_ptr(), lib.rihip_inbatch_loss_parts(B, B), 1.0 / (B * (B - 1.0)),
                                   loss.data_ptr(), st), "sum_partials")
    return loss, dU, dI


class _InBatchFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, I):
        if U.shape[0] < 2:  # reference: mean over an empty negative set -> nan (two_tower.py:156-158)
            ctx.save_for_backward(torch.zeros_like(U), torch.zeros_like(I))
            ctx.in_devs = (U.device, I.device)
            return torch.full((), float("nan"), dtype=torch.float32, device=U.device)
        loss, dU, dI = inbatch_loss_and_grads(U, I)
        ctx.save_for_backward(dU, dI)
        ctx.in_devs = (U.device, I.device)
        return loss

    @staticmethod
    def backward(ctx, g):
        dU, dI = ctx.saved_tensors
        g = g.to(dU.device)
        return (dU * g).to(ctx.in_devs[0]), (dI * g).to(ctx.in_devs[1])


class TwoTowerModel(nn.Module):
    """two_tower.py:75-251 with HIP towers / losses."""

    def __init__(self, n_users: int, n_items: int, embed_dim: int = 64, hidden_dim: int = 128, dropout: float = 0.1):
        super().__init__()
        self.n_users = int(n_users)
        self.n_items = int(n_items)
        self.embed_dim = int(embed_dim)
        self.hidden_dim = int(hidden_dim)
        dev = L.device() if L.have_gpu() else None   # parameters are created directly on the HIP device
        self.user_tower = UserTower(self.n_users, embed_dim, hidden_dim, dropout, device=dev)
        self.item_tower = ItemTower(self.n_items, embed_dim, hidden_dim, dropout, device=dev)
        self._item_embeddings: Optional[torch.Tensor] = None
        self._item_id_to_idx: Optional[Dict[int, int]] = None
        self._idx_to_item_id: Optional[Dict[int, int]] = None

    # -- forward / losses -------------------------------------------------------------------
    def forward(self, user_ids, pos_item_ids, pos_genre_vectors, neg_item_ids=None, neg_genre_vectors=None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        """two_tower.py:102-115: returns (user_emb, pos_item_emb); neg_* are ignored like the reference."""
        return self.user_tower(user_ids), self.item_tower(pos_item_ids, pos_genre_vectors)

    def bpr_loss(self, user_emb, pos_item_emb, neg_item_emb) -> torch.Tensor:
        """two_tower.py:117-130."""
        in_dev = user_emb.device
        loss = _BprPairFn.apply(user_emb, pos_item_emb, neg_item_emb)
        return loss if in_dev.type == "cuda" else loss.to(in_dev)

    def in_batch_bpr_loss(self, user_emb, item_emb) -> torch.Tensor:
        """two_tower.py:132-160 (closed form of the per-row python loop)."""
        in_dev = user_emb.device
        loss = _InBatchFn.apply(user_emb, item_emb)
        return loss if in_dev.type == "cuda" else loss.to(in_dev)

    # -- inference helpers (two_tower.py:166-210) ---------------------------------------------
    @torch.no_grad()
    def get_user_embedding(self, user_id: int, device: torch.device = torch.device("cpu")) -> np.ndarray:
        self.eval()
        uid = torch.tensor([user_id], dtype=torch.long, device=L.device())
        return self.user_tower(uid).cpu().numpy()[0]

    @torch.no_grad()
    def get_user_embeddings(self, user_ids, as_tensor: bool = False):
        """Batched variant (not in the reference): [n] ids -> [n,d]; stays on device if as_tensor."""
        self.eval()
        uid = torch.as_tensor(user_ids, dtype=torch.long, device=L.device())
        out = self.user_tower(uid)
        return out if as_tensor else out.cpu().numpy()

    @torch.no_grad()
    def get_item_embeddings(self, item_ids: List[int], genre_vectors: np.ndarray,
                            device: torch.device = torch.device("cpu"), batch_size: int = 512) -> np.ndarray:
        self.eval()
        dev = L.device()
        ids = torch.as_tensor(np.asarray(item_ids), dtype=torch.long, device=dev)
        g = torch.as_tensor(np.asarray(genre_vectors), dtype=torch.float32, device=dev)
        outs = []
        for s in range(0, len(item_ids), batch_size):  # same 512-row batching as the reference (:188)
            outs.append(self.item_tower(ids[s:s + batch_size], g[s:s + batch_size]))
        return torch.cat(outs, 0).cpu().numpy() if outs else np.zeros((0, self.embed_dim), np.float32)

    def precompute_item_embeddings(self, item_ids: List[int], genre_vectors: np.ndarray,
                                   device: torch.device = torch.device("cpu")) -> None:
        embs = self.get_item_embeddings(item_ids, genre_vectors, device)
        self._item_embeddings = torch.tensor(embs, dtype=torch.float32)
        self._item_id_to_idx = {iid: idx for idx, iid in enumerate(item_ids)}
        self._idx_to_item_id = {idx: iid for idx, iid in enumerate(item_ids)}

    # -- persistence (two_tower.py:216-251; same checkpoint keys) -----------------------------
    def save(self, path: str) -> None:
        save_path = Path(path)
        save_path.parent.mkdir(parents=True, exist_ok=True)
        torch.save(
            {
                "state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()},
                "n_users": self.n_users,
                "n_items": self.n_items,
                "embed_dim": self.embed_dim,
                "item_id_to_idx": self._item_id_to_idx,
                "idx_to_item_id": self._idx_to_item_id,
            },
            save_path,
        )
        logger.info("Saved two-tower model to %s", save_path)

    @classmethod
    def load(cls, path: str, device: torch.device = torch.device("cpu")) -> "TwoTowerModel":
        """Reads reference checkpoints and its own.  hidden_dim is inferred from the weights
        (the reference hard-codes 128 and cannot re-load other sizes: SURVEY.md Appendix B)."""
        checkpoint = torch.load(path, map_location="cpu", weights_only=True)
        sd = checkpoint["state_dict"]
        hidden = int(sd["user_tower.mlp.0.weight"].shape[0])
        model = cls(n_users=int(checkpoint["n_users"]), n_items=int(checkpoint["n_items"]),
                    embed_dim=int(checkpoint["embed_dim"]), hidden_dim=hidden)
        model.load_state_dict(sd)
        model._item_id_to_idx = checkpoint.get("item_id_to_idx")
        model._idx_to_item_id = checkpoint.get("idx_to_item_id")
        if L.have_gpu():
            model.to(L.device())
        model.eval()
        logger.info("Loaded two-tower model from %s (users=%d, items=%d, dim=%d)", path, model.n_users,
                    model.n_items, model.embed_dim)
        return model
