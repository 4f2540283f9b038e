"""GPU-resident serving stage: user tower -> inner-product retrieval -> feature assembly -> LambdaMART.

Mirrors the online chain of the reference's RecommendationPipeline.get_recommendations
(src/serving/recommender.py:269-387): A11 get_user_embedding -> R2 search(k=500) -> feature fetch
(:319-322) -> S1 _build_ranking_features (:213-263) -> K1 ranker.predict (:338) -> nlargest(k) (:346),
but for a BATCH of users and without leaving the device between stages.  The reference class itself stays
the caller for single requests (drop-in through the three model classes); this module is the "next" row
§8f-1 of SURVEY.md: a GPU feature table that supersedes the Redis MGET + the 500-iteration Python loop.
"""
from __future__ import annotations

import os

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .faiss_index import FAISSIndex
from .ranker import LightGBMRanker
from .two_tower import N_GENRES, TwoTowerModel

USER_SCALARS = [("avg_rating", 3.5), ("log_rating_count", 0.0), ("recency_score", 0.5), ("gender_encoded", 0.0),
                ("age_normalized", 0.3), ("occupation_normalized", 0.3)]            # recommender.py:227-232
ITEM_SCALARS = [("avg_rating", 3.5), ("log_rating_count", 0.0), ("popularity_score", 0.0), ("rating_stddev", 0.0),
                ("year_normalized", 0.5)]                                           # recommender.py:234-238


def feature_columns() -> List[str]:
    """The 50 ranking columns in the order of the reference's get_feature_columns
    (src/features/feature_engineering.py:434-443)."""
    return (["avg_rating", "log_rating_count", "recency_score", "gender_encoded", "age_normalized",
             "occupation_normalized", "item_avg_rating", "item_log_rating_count", "popularity_score", "rating_stddev",
             "year_normalized", "rating_diff", "user_item_popularity_ratio", "genre_affinity"]
            + [f"user_genre_{i}" for i in range(N_GENRES)] + [f"item_genre_{i}" for i in range(N_GENRES)])


class GpuFeatureStore:
    """float64 feature tables on the device; rows not loaded keep the reference's defaults.
    Replaces RedisFeatureStore.get_user_features / get_item_features_batch (src/features/feature_store.py:113-150)
    on the accelerated path (same keys inside the per-entity dicts)."""

    def __init__(self, n_users: int, n_items: int):
        UW, IW = 6 + N_GENRES, 5 + N_GENRES
        self.user = np.zeros((n_users + 1, UW), dtype=np.float64)
        self.item = np.zeros((n_items + 1, IW), dtype=np.float64)
        self.user[:, :6] = [d for _, d in USER_SCALARS]
        self.item[:, :5] = [d for _, d in ITEM_SCALARS]
        self._dev: Optional[Tuple[torch.Tensor, torch.Tensor]] = None

    def set_user_features(self, user_id: int, feat: Dict[str, Any]) -> None:
        row = self.user[user_id]
        for j, (name, dflt) in enumerate(USER_SCALARS):
            row[j] = float(feat.get(name, dflt))
        gp = feat.get("genre_pref", [0.0] * N_GENRES)
        for i in range(N_GENRES):
            row[6 + i] = float(gp[i]) if i < len(gp) else 0.0
        self._dev = None

    def set_item_features(self, item_id: int, feat: Optional[Dict[str, Any]]) -> None:
        feat = feat or {}
        row = self.item[item_id]
        for j, (name, dflt) in enumerate(ITEM_SCALARS):
            row[j] = float(feat.get(name, dflt))
        gv = feat.get("genre_vector", [0.0] * N_GENRES)
        for i in range(N_GENRES):
            row[5 + i] = float(gv[i]) if i < len(gv) else 0.0
        self._dev = None

    def load_arrays(self, user_tab: Optional[np.ndarray] = None, item_tab: Optional[np.ndarray] = None) -> None:
        """Bulk load (e.g. from the parquet feature files): arrays in the table layout, row index = id."""
        if user_tab is not None:
            self.user[: len(user_tab)] = user_tab
        if item_tab is not None:
            self.item[: len(item_tab)] = item_tab
        self._dev = None

    def load_all_features(self, user_features_df, item_features_df, batch_size: int = 500) -> None:
        """Bulk load from the DataFrames the reference's FeatureEngineer produces / saves (same entry point and column
        conventions as RedisFeatureStore.load_all_features, src/features/feature_store.py:156-228): scalar columns by
        name, `genre_pref_<i>` / `genre_vec_<i>` expanded columns (feature_engineering.py:382-404) or the un-expanded
        `genre_pref` / `genre_vector` array columns; columns that are absent keep the defaults of
        _build_ranking_features (recommender.py:227-238); ids beyond the table size grow it.  Vectorised: no per-row
        Python loop (the reference iterates rows and msgpack-serialises each one)."""
        def fill(tab, df, id_col, scalars, vec_prefix, vec_col, off):
            if df is None or len(df) == 0:
                return tab
            ids = df[id_col].to_numpy().astype(np.int64)
            if ids.min() < 0:
                raise ValueError(f"negative {id_col}")
            if ids.max() >= tab.shape[0]:          # grow, new rows at their defaults
                grown = np.repeat(tab[:1].copy(), int(ids.max()) + 1, axis=0)
                grown[:, :] = self._default_row(tab.shape[1], scalars)
                grown[: tab.shape[0]] = tab
                tab = grown
            for j, (name, _) in enumerate(scalars):
                if name in df.columns:
                    tab[ids, j] = df[name].to_numpy().astype(np.float64)
            vec_cols = [f"{vec_prefix}{i}" for i in range(N_GENRES)]
            if all(c in df.columns for c in vec_cols):
                tab[ids, off:off + N_GENRES] = df[vec_cols].to_numpy().astype(np.float64)
            elif vec_col in df.columns:
                tab[ids, off:off + N_GENRES] = np.stack([np.asarray(v, dtype=np.float64)[:N_GENRES]
                                                         for v in df[vec_col].to_numpy()])
            return tab

        self.user = fill(self.user, user_features_df, "user_id", USER_SCALARS, "genre_pref_", "genre_pref", 6)
        self.item = fill(self.item, item_features_df, "item_id", ITEM_SCALARS, "genre_vec_", "genre_vector", 5)
        self._dev = None

    @staticmethod
    def _default_row(width: int, scalars) -> np.ndarray:
        row = np.zeros((width,), dtype=np.float64)
        row[: len(scalars)] = [d for _, d in scalars]
        return row

    @classmethod
    def from_parquet(cls, features_dir: str, n_users: int = 0, n_items: int = 0) -> "GpuFeatureStore":
        """Device feature tables straight from `user_features.parquet` / `item_features.parquet` as written by
        FeatureEngineer.save_features (src/features/feature_engineering.py:376-406) -- the parquet -> device-table
        loader SURVEY.md §8f-1 names.  A missing file leaves that table at its defaults (the reference's
        load_features skips missing files too, :415, :425)."""
        import pandas as pd
        from pathlib import Path
        d = Path(features_dir)
        up, ip = d / "user_features.parquet", d / "item_features.parquet"
        udf = pd.read_parquet(up) if up.exists() else None
        idf = pd.read_parquet(ip) if ip.exists() else None
        nu = max(n_users, int(udf["user_id"].max()) if udf is not None and len(udf) else 0)
        ni = max(n_items, int(idf["item_id"].max()) if idf is not None and len(idf) else 0)
        st = cls(nu, ni)
        st.load_all_features(udf, idf)
        return st

    def device_tables(self) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._dev is None:
            dev = L.device()
            self._dev = (torch.from_numpy(self.user).to(dev), torch.from_numpy(self.item).to(dev))
        return self._dev


def build_ranking_features_device(store: GpuFeatureStore, user_ids: torch.Tensor, cand_ids: torch.Tensor,
                                  feature_names: Sequence[str]) -> torch.Tensor:
    """X f32 [nq*kc, len(feature_names)] on device (the matrix ranker.predict would see)."""
    lib = L.lib()
    ut, it = store.device_tables()
    key = tuple(feature_names)
    col_map = store._col_maps.get(key) if hasattr(store, "_col_maps") else None
    if col_map is None or col_map.device != ut.device:   # one H2D copy per feature list, not per request
        canon = {n: i for i, n in enumerate(feature_columns())}
        col_map = torch.tensor([canon.get(n, -1) for n in feature_names], dtype=torch.int32, device=ut.device)
        if not hasattr(store, "_col_maps"):
            store._col_maps = {}
        store._col_maps[key] = col_map
    uid = L.i64c(user_ids)
    cand = L.i64c(cand_ids)
    nq, kc = cand.shape
    X = torch.empty((nq * kc, len(feature_names)), dtype=torch.float32, device=ut.device)
    L.check(lib.rihip_rank_features_build(ut.data_ptr(), ut.shape[0], it.data_ptr(), it.shape[0], uid.data_ptr(),
                                          cand.data_ptr(), nq, kc, col_map.data_ptr(), len(feature_names),
                                          X.data_ptr(), L.stream_ptr()), "rank_features_build")
    return X


class GpuRecommendationPipeline:
    def __init__(self, model: TwoTowerModel, index: FAISSIndex, ranker: LightGBMRanker, store: GpuFeatureStore,
                 top_k_candidates: int = 500, top_k_results: int = 20):
        """defaults = settings.TOP_K_CANDIDATES / TOP_K_RESULTS (src/config.py:11-12)"""
        self.model, self.index, self.ranker, self.store = model, index, ranker, store
        self.top_k_candidates, self.top_k_results = top_k_candidates, top_k_results
        self._graphs: Dict[Tuple[int, int], Any] = {}
        self._pin: Dict[int, Any] = {}
        self._defer = os.environ.get("RIHIP_SERVE_DEFER", "1") != "0"   # 0: exactness check inside the search (experiments)

    @torch.no_grad()
    def recommend_batch(self, user_ids, k: Optional[int] = None, graph: bool = False
                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """-> (item_ids i64 [nq,k], ranker scores f64 [nq,k], retrieval scores f32 [nq,k]) on device; -1 padded
        where retrieval returned fewer than k candidates.  Ties in the ranker score keep retrieval order
        (DataFrame.nlargest(keep='first'), recommender.py:346).

        graph=True (small request batches): the ~20 launches of the chain are captured once per (batch size, k) into a
        hipGraph and replayed -- a single request is launch-bound otherwise.  The returned tensors are the graph's static
        outputs: valid until the next replay of the same shape.  Falls back to the eager chain when the shape takes a
        path with a host synchronisation inside the chain."""
        k = k or self.top_k_results
        # The retrieval stage's exactness check is deferred to the END of the chain (FAISSIndex.set_deferred_check): the
        # thresholded IVF pass of a large batch used to stop for a host round trip in the middle of the chain (a 54 us
        # hole at 256 requests, and the reason such batches could not be captured as a hipGraph).  In the rare case that
        # queries had to be re-done exactly, the chain runs again on the corrected candidates (not deferred).
        if not self._defer:
            out = self._replay(user_ids, k) if graph else None
            if out is not None:
                return out[0]
            return self._chain(self._ids_to_device(user_ids), k)
        self.index.set_deferred_check(True)
        try:
            out = self._replay(user_ids, k) if graph else None
            uid = None
            if out is not None:
                out, redone = out
            else:
                uid = self._ids_to_device(user_ids)
                out = self._chain(uid, k)
                redone = self.index.finish_search()
        finally:
            self.index.set_deferred_check(False)
        if redone:
            if uid is None:
                uid = self._ids_to_device(user_ids)
            out = self._chain(uid, k)
        return out

    def _ids_to_device(self, user_ids) -> torch.Tensor:
        """user ids -> device without stalling the host: `torch.as_tensor(list, device=...)` is a pageable copy, which
        blocks the host until everything already enqueued on the stream (the previous batch's chain) has run -- a 0.1 ms
        hole per 256-request batch.  A pinned staging buffer per batch size + an asynchronous copy lets the host enqueue
        batch k+1 while the GPU still works on batch k."""
        if isinstance(user_ids, torch.Tensor) and user_ids.is_cuda:
            return user_ids.to(dtype=torch.long)
        n = len(user_ids)
        ent = self._pin.get(n)
        if ent is None:
            ent = (torch.empty((n,), dtype=torch.long).pin_memory(), torch.empty((n,), dtype=torch.long, device=L.device()),
                   torch.cuda.Event())
            self._pin[n] = ent
        pin, dev_t, ev = ent
        ev.synchronize()                        # the previous copy out of the staging buffer has executed
        pin.copy_(torch.as_tensor(user_ids, dtype=torch.long))
        dev_t.copy_(pin, non_blocking=True)
        ev.record()
        return dev_t

    def _graph_state(self):
        """everything a captured chain bakes in besides the torch-owned tensors of its own pool: the library's scratch
        generation (handle-owned buffers that are freed when they grow, nprobe, id map, index / forest content) and the
        identity of the feature tables and of the three stage objects"""
        ut, it = self.store.device_tables()
        return (int(L.lib().rihip_scratch_generation()), ut.data_ptr(), it.data_ptr(), id(self.index), id(self.ranker),
                id(self.model), self.top_k_candidates, tuple(self.ranker.feature_names))

    def _replay(self, user_ids, k: int):
        nq = len(user_ids)
        key = (nq, k)
        ent = self._graphs.get(key)
        if ent is not None and ent is not False and ent[3] != self._graph_state():
            # an eager call (or a capture of a larger shape) grew a scratch buffer, nprobe changed, the feature tables
            # were reloaded, ...: the pointers inside this graph are stale -- drop it and capture again
            ent = None
            del self._graphs[key]
        if ent is None:
            dev = L.device()
            su = torch.ones((nq,), dtype=torch.long, device=dev)
            try:
                cur = torch.cuda.current_stream(dev)
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):      # warm-up: scratch buffers, LDS grants, lazy module loads
                    for _ in range(2):
                        self._chain(su, k)
                        self.index.finish_search()
                cur.wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = self._chain(su, k)
                deferred = self.index.search_pending()      # the captured search left its exactness check to the caller
                self.index.set_deferred_check(True)         # (capture ran nothing: drop the pending state)
                ent = (g, su, out, self._graph_state(), deferred)   # state recorded AFTER capture: the warm-up may have grown scratch
            except Exception:                        # a path with a host sync cannot be captured: stay eager for this shape
                torch.cuda.synchronize()
                ent = False
            self._graphs[key] = ent
        if ent is False:
            return None
        g, su, out, _, deferred = ent
        su.copy_(torch.as_tensor(user_ids, dtype=torch.long), non_blocking=True)
        g.replay()
        # a replay runs no host code: the failure count of its deferred search is read here (one synchronisation)
        return out, (self.index.last_fail_count() if deferred else 0)

    def _chain(self, uid: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        q = self.model.get_user_embeddings(uid, as_tensor=True)
        # tower outputs are already L2-normalised (two_tower.py:42): the wrapper's re-normalisation (faiss_index.py:108-110)
        # would divide by 1 +- 1e-7 and cost three tensor ops per request
        rs, cand = self.index.batch_search_device(q, k=self.top_k_candidates, normalized=True)
        nq, kc = cand.shape
        X = build_ranking_features_device(self.store, uid, cand, self.ranker.feature_names)
        scores = self.ranker.predict_device(X)
        k = min(k, kc)
        ids = torch.empty((nq, k), dtype=torch.int64, device=cand.device)
        top = torch.empty((nq, k), dtype=torch.float64, device=cand.device)
        trs = torch.empty((nq, k), dtype=torch.float32, device=cand.device)
        L.check(L.lib().rihip_rank_topk(scores.data_ptr(), cand.data_ptr(), rs.data_ptr(), nq, kc, k, ids.data_ptr(),
                                        top.data_ptr(), trs.data_ptr(), L.stream_ptr()), "rank_topk")
        return ids, top, trs

    def get_recommendations(self, user_id: int, k: Optional[int] = None, graph: bool = False) -> List[Dict[str, Any]]:
        ids, sc, rs = self.recommend_batch([user_id], k, graph=graph)
        out = []
        for rank, (i, s, r) in enumerate(zip(ids[0].tolist(), sc[0].tolist(), rs[0].tolist()), start=1):
            if i >= 0:
                out.append({"item_id": int(i), "score": float(s), "rank": rank, "retrieval_score": float(r)})
        return out
