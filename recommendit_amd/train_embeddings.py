"""Mirror of the reference's src/training/train_embeddings.py for the MI355X path.

``UserItemDataset`` keeps the reference constructor and per-sample ``__getitem__`` (:23-79) and
adds a device-side batch sampler (positives shuffled on the GPU, one rejection-sampled negative per
positive drawn by a HIP kernel from the catalogue and rejected while in the user's *rated* set
-- :58-63 --, genre rows gathered from a device table), because the reference's own DataLoader tops out at ~21 k
samples/s and would starve the kernels.

``EmbeddingTrainer`` keeps the reference's constructor arguments and ``train()`` flow (:131-223:
Adam(lr, wd=1e-5) + clip 1.0 + CosineAnnealingLR stepped per epoch, best-loss checkpoint, final
precompute + save) and drives ``HipBPRTrainer`` instead of autograd.
"""
from __future__ import annotations

import logging
import time
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import pandas as pd
import torch

from . import _lib as L
from .synthetic import GENRES, N_GENRES
from .trainer import HipBPRTrainer, cosine_lr
from .two_tower import TwoTowerModel

logger = logging.getLogger(__name__)
GENRE_TO_IDX = {g: i for i, g in enumerate(GENRES)}


def load_ml1m(data_dir: str) -> Tuple[pd.DataFrame, pd.DataFrame, pd.DataFrame]:
    """ratings/users/movies in the '::' format (reference src/features/feature_engineering.py:39-72)."""
    d = Path(data_dir)
    for sub in ("", "ml-1m"):
        if (d / sub / "ratings.dat").exists():
            d = d / sub
            break
    ratings = pd.read_csv(d / "ratings.dat", sep="::", engine="python", header=None,
                          names=["user_id", "item_id", "rating", "timestamp"], encoding="latin-1")
    users = pd.read_csv(d / "users.dat", sep="::", engine="python", header=None,
                        names=["user_id", "gender", "age", "occupation", "zip_code"], encoding="latin-1")
    movies = pd.read_csv(d / "movies.dat", sep="::", engine="python", header=None,
                         names=["item_id", "title", "genres"], encoding="latin-1")
    return ratings, users, movies


def build_item_genre_dict(movies_df: pd.DataFrame) -> Dict[int, np.ndarray]:
    """train_embeddings.py:118-129."""
    result = {}
    for iid, genres in zip(movies_df["item_id"].values, movies_df["genres"].values):
        vec = np.zeros(N_GENRES, dtype=np.float32)
        for g in str(genres).split("|"):
            idx = GENRE_TO_IDX.get(g)
            if idx is not None:
                vec[idx] = 1.0
        result[int(iid)] = vec
    return result


class UserItemDataset:
    def __init__(self, ratings_df: pd.DataFrame, item_genre_dict: Dict[int, np.ndarray], all_item_ids: List[int],
                 n_negatives: int = 4, min_rating: float = 4.0, device_tables: bool = True):
        self.item_genre_dict = item_genre_dict
        self.all_item_ids = all_item_ids
        self.n_negatives = n_negatives  # accepted and unused, like the reference (:35,:40)
        positives = ratings_df[ratings_df["rating"] >= min_rating][["user_id", "item_id"]]
        self.user_ids = positives["user_id"].values.astype(np.int64)
        self.item_ids = positives["item_id"].values.astype(np.int64)
        self.all_items_array = np.array(all_item_ids, dtype=np.int64)
        self._ratings_u = ratings_df["user_id"].values.astype(np.int64)
        self._ratings_i = ratings_df["item_id"].values.astype(np.int64)
        self._user_rated: Optional[Dict[int, set]] = None
        self._dev_ready = False
        logger.info("Dataset: %d positive pairs", len(self.user_ids))

    def __len__(self) -> int:
        return len(self.user_ids)

    # ---- reference-compatible host path (train_embeddings.py:58-79) ---------------------------
    @property
    def user_rated(self) -> Dict[int, set]:
        if self._user_rated is None:
            df = pd.DataFrame({"u": self._ratings_u, "i": self._ratings_i})
            self._user_rated = df.groupby("u")["i"].apply(set).to_dict()
        return self._user_rated

    def _sample_negative(self, user_id: int) -> int:
        rated = self.user_rated.get(user_id, set())
        while True:
            neg_id = int(np.random.choice(self.all_items_array))
            if neg_id not in rated:
                return neg_id

    def __getitem__(self, idx: int):
        user_id = int(self.user_ids[idx])
        pos = int(self.item_ids[idx])
        neg = self._sample_negative(user_id)
        zero = np.zeros(N_GENRES, dtype=np.float32)
        return (torch.tensor(user_id, dtype=torch.long), torch.tensor(pos, dtype=torch.long),
                torch.tensor(self.item_genre_dict.get(pos, zero), dtype=torch.float32),
                torch.tensor(neg, dtype=torch.long),
                torch.tensor(self.item_genre_dict.get(neg, zero), dtype=torch.float32))

    # ---- device-side sampler ---------------------------------------------------------------------
    def _prepare_device(self) -> None:
        if self._dev_ready:
            return
        dev = L.device()
        self.M = int(max(self._ratings_i.max(), self.all_items_array.max())) + 1
        self.d_pos_u = torch.from_numpy(self.user_ids).to(dev)
        self.d_pos_i = torch.from_numpy(self.item_ids).to(dev)
        keys = np.unique(self._ratings_u * self.M + self._ratings_i)
        self.d_rated = torch.from_numpy(keys).to(dev)               # sorted (user*M + item) of every rating
        self.d_catalog = torch.from_numpy(self.all_items_array).to(dev)
        gm = np.zeros((self.M, N_GENRES), dtype=np.float32)          # ids absent from movies.dat -> zeros (:70-71)
        for iid, vec in self.item_genre_dict.items():
            if 0 <= iid < self.M:
                gm[iid] = vec
        self.d_genres = torch.from_numpy(gm).to(dev)
        self._dev_ready = True

    def sample_negatives(self, users: torch.Tensor, gen: torch.Generator, max_attempts: int = 1000) -> torch.Tensor:
        """One HIP launch for the whole batch (rihip_sample_negatives): uniform catalogue draws, re-drawn while the
        item is in the user's rated set -- the reference's rejection loop (:58-63), bounded at `max_attempts`."""
        self._prepare_device()
        u = users.contiguous()
        neg = torch.empty_like(u)
        self._neg_calls = getattr(self, "_neg_calls", 0) + 1     # host-side counter: no device sync for the seed
        seed = (gen.initial_seed() * 0x9E3779B97F4A7C15 + self._neg_calls) & ((1 << 63) - 1)
        L.check(L.lib().rihip_sample_negatives(u.data_ptr(), u.numel(), self.d_catalog.data_ptr(),
                                               self.d_catalog.numel(), self.d_rated.data_ptr(), self.d_rated.numel(),
                                               self.M, seed, max_attempts, neg.data_ptr(), None, L.stream_ptr()),
                "sample_negatives")
        return neg

    def epoch_batches(self, batch_size: int, gen: torch.Generator, with_negatives: bool = True
                      ) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """Shuffled, drop_last batches (DataLoader(shuffle=True, drop_last=True), :144-151) as device
        tensors: (user_ids [B], item_ids [2B]=pos||neg or [B], item_genres)."""
        self._prepare_device()
        n = len(self)
        perm = torch.randperm(n, device=self.d_pos_u.device, generator=gen)
        for s in range(0, n - batch_size + 1, batch_size):
            idx = perm[s:s + batch_size]
            u, p = self.d_pos_u[idx], self.d_pos_i[idx]
            if with_negatives:
                items = torch.cat([p, self.sample_negatives(u, gen)])
            else:
                items = p
            yield u, items, self.d_genres[items]


class EmbeddingTrainer:
    def __init__(self, data_dir: str = "data/ml-1m", model_output_path: str = "models/two_tower.pt",
                 embed_dim: int = 64, epochs: int = 10, batch_size: int = 1024, learning_rate: float = 1e-3,
                 device: Optional[str] = None, loss_mode: str = "sampled", table_opt: str = "dense",
                 dropout: float = 0.1, seed: int = 0):
        """Defaults = the reference's settings (src/config.py:13,:24-26); `device` is accepted for
        signature compatibility -- training always runs on the HIP device."""
        self.data_dir, self.model_output_path = data_dir, model_output_path
        self.embed_dim, self.epochs, self.batch_size, self.learning_rate = embed_dim, epochs, batch_size, learning_rate
        self.loss_mode, self.table_opt, self.dropout, self.seed = loss_mode, table_opt, dropout, seed
        self.device = L.device()
        self.history: List[Dict] = []

    def load_data(self):
        return load_ml1m(self.data_dir)

    def train(self, ratings_df: Optional[pd.DataFrame] = None, movies_df: Optional[pd.DataFrame] = None
              ) -> TwoTowerModel:
        if ratings_df is None:
            ratings_df, _, movies_df = self.load_data()
        n_users = int(ratings_df["user_id"].max())
        n_items = int(ratings_df["item_id"].max())
        all_item_ids = sorted(movies_df["item_id"].unique().tolist())
        item_genre_dict = build_item_genre_dict(movies_df)
        dataset = UserItemDataset(ratings_df, item_genre_dict, all_item_ids)
        torch.manual_seed(self.seed)
        model = TwoTowerModel(n_users, n_items, embed_dim=self.embed_dim, hidden_dim=128, dropout=self.dropout)
        trainer = HipBPRTrainer(model, self.batch_size, lr=self.learning_rate, weight_decay=1e-5, max_norm=1.0,
                                loss_mode=self.loss_mode, table_opt=self.table_opt, seed=self.seed)
        gen = torch.Generator(device=self.device)
        gen.manual_seed(self.seed)
        Path(self.model_output_path).parent.mkdir(parents=True, exist_ok=True)
        best = float("inf")
        for epoch in range(1, self.epochs + 1):
            model.train()
            lr = cosine_lr(self.learning_rate, epoch - 1, self.epochs)
            t0 = time.time()
            tot = torch.zeros((), dtype=torch.float64, device=self.device)
            nb = 0
            for u, items, genres in dataset.epoch_batches(self.batch_size, gen, self.loss_mode == "sampled"):
                tot += trainer.step(u, items, genres, lr=lr).double()
                nb += 1
            avg = float(tot.item()) / max(nb, 1)   # one host sync per epoch (the reference syncs every step, :194)
            dt = time.time() - t0
            self.history.append({"epoch": epoch, "loss": avg, "seconds": dt, "pairs_per_sec": nb * self.batch_size / dt})
            logger.info("Epoch %d/%d - loss %.4f - %.1fs - lr %.6f", epoch, self.epochs, avg, dt, lr)
            if avg < best:
                best = avg
                model.save(self.model_output_path)
        genre_matrix = np.stack([item_genre_dict.get(i, np.zeros(N_GENRES, np.float32)) for i in all_item_ids])
        model.precompute_item_embeddings(all_item_ids, genre_matrix, self.device)
        model.save(self.model_output_path)
        self.trainer = trainer
        return model


def retrieval_ndcg(model: TwoTowerModel, ratings_df: pd.DataFrame, movies_df: pd.DataFrame, k_candidates: int = 500,
                   n_eval_users: int = 200, exact: bool = True) -> Dict[str, float]:
    """The reference's `run_evaluate` protocol (src/pipelines/run_pipeline.py:153-230) in its Redis-less form
    (identical default features -> ranker scores tie -> retrieval order decides):
    test set = last N ratings per user by timestamp, N = max(1, int(len*0.1/n_users)); first 200 users;
    ground truth = test items rated >= 4; candidates = top-500 by inner product; NDCG@{5,10,20} of the top 20."""
    from .faiss_index import FAISSIndex
    from .metrics import evaluate_model
    n_users = ratings_df["user_id"].nunique()
    n_test = max(1, int(len(ratings_df) * 0.1 / n_users))
    test = ratings_df.sort_values("timestamp").groupby("user_id").tail(n_test)
    eval_users = test["user_id"].unique()[:n_eval_users]
    item_ids = sorted(movies_df["item_id"].unique().tolist())
    gd = build_item_genre_dict(movies_df)
    gm = np.stack([gd.get(i, np.zeros(N_GENRES, np.float32)) for i in item_ids])
    embs = model.get_item_embeddings(item_ids, gm)
    index = FAISSIndex(embed_dim=model.embed_dim, exact=exact)
    index.build_ivf_index(embs.astype(np.float32), item_ids)
    U = model.get_user_embeddings(eval_users.astype(np.int64))
    _, ids = index.batch_search(U, k=k_candidates)
    truth = {int(u): g[g["rating"] >= 4]["item_id"].tolist() for u, g in test[test["user_id"].isin(eval_users)].groupby("user_id")}
    recs = {int(u): [int(x) for x in ids[i][:20] if x >= 0] for i, u in enumerate(eval_users)}
    return evaluate_model(recs, truth, [5, 10, 20])
