"""Batched offline evaluation -- the reference's ``PipelineRunner.run_evaluate`` protocol
(src/pipelines/run_pipeline.py:121-237) over the device-resident serving chain.

The reference walks the evaluation users one at a time: user tower -> FAISS search(500) -> per-candidate feature dicts
-> DataFrame -> ranker.predict -> nlargest(20) (:166-220).  Here the same protocol is ONE call per batch of users
through GpuRecommendationPipeline.recommend_batch; the split, the ground truth and the metric definitions are the
reference's: test set = the last ``max(1, int(len(ratings)*0.1/n_users))`` ratings of each user by timestamp (:153-157),
the first ``n_eval_users`` users of that set (:160), ground truth = their test items rated >= 4 (:171-173), users
without ground truth are skipped (:174-175), top-20 by ranker score, NDCG/Recall@{5,10,20} + MRR (+ catalog coverage)
from evaluate_model (:222-227).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np

from .metrics import evaluate_model
from .recommender import GpuRecommendationPipeline


def run_evaluate(pipe: GpuRecommendationPipeline, ratings_df, movies_df=None, n_eval_users: Optional[int] = 200,
                 top_k: int = 20, batch_size: int = 256, k_values: Optional[List[int]] = None) -> Dict[str, Any]:
    ratings = ratings_df.sort_values("timestamp")
    n_test = max(1, int(len(ratings) * 0.1 / ratings["user_id"].nunique()))
    test = ratings.groupby("user_id").tail(n_test)
    eval_users = test["user_id"].unique()
    if n_eval_users:
        eval_users = eval_users[:n_eval_users]
    sub = test[test["user_id"].isin(eval_users)]
    truth = {int(u): g[g["rating"] >= 4]["item_id"].tolist() for u, g in sub.groupby("user_id")}
    users = [int(u) for u in eval_users if truth.get(int(u))]
    recs: Dict[int, List[int]] = {}
    for s in range(0, len(users), batch_size):
        chunk = users[s:s + batch_size]
        ids, _, _ = pipe.recommend_batch(chunk, k=top_k)
        ids = ids.cpu().numpy()
        for u, row in zip(chunk, ids):
            recs[u] = [int(x) for x in row if x >= 0]
    res = evaluate_model(recs, truth, k_values or [5, 10, 20])
    if movies_df is not None:     # catalog coverage (evaluate_model's catalog_size argument, run_pipeline.py:226)
        shown = {i for r in recs.values() for i in r}
        res["coverage"] = len(shown) / max(1, int(movies_df["item_id"].nunique()))
    res["n_eval_users"] = len(users)
    return res
