"""Ranking metrics used by the parity harness -- same definitions as the reference's
src/evaluation/metrics.py (ndcg_at_k :20-69 binary relevance, recall_at_k :72-87, mrr :104-118,
evaluate_model :301-384: mean over users that have at least one relevant item)."""
from __future__ import annotations

import math
from typing import Any, Dict, List, Sequence


def ndcg_at_k(recommended: Sequence[Any], relevant: Sequence[Any], k: int) -> float:
    relevant_set = set(relevant)
    dcg = 0.0
    for i, item in enumerate(list(recommended)[:k]):
        if item in relevant_set:
            dcg += 1.0 / math.log2(i + 2)
    idcg = sum(1.0 / math.log2(i + 2) for i in range(min(len(relevant), k)))
    return 0.0 if idcg == 0 else dcg / idcg


def recall_at_k(recommended: Sequence[Any], relevant: Sequence[Any], k: int) -> float:
    if not relevant:
        return 0.0
    relevant_set = set(relevant)
    return sum(1 for item in list(recommended)[:k] if item in relevant_set) / len(relevant_set)


def mrr(recommended: Sequence[Any], relevant: Sequence[Any]) -> float:
    relevant_set = set(relevant)
    for rank, item in enumerate(recommended, start=1):
        if item in relevant_set:
            return 1.0 / rank
    return 0.0


def evaluate_model(recommendations_by_user: Dict[Any, List[Any]], ground_truth_by_user: Dict[Any, List[Any]],
                   k_values: List[int] = None) -> Dict[str, Any]:
    k_values = k_values or [5, 10, 20]
    res: Dict[str, Any] = {"n_users": len(recommendations_by_user), "k_values": k_values}
    per = {k: {"ndcg": [], "recall": []} for k in k_values}
    mrrs = []
    for u, recs in recommendations_by_user.items():
        rel = ground_truth_by_user.get(u, [])
        if not rel:
            continue
        for k in k_values:
            per[k]["ndcg"].append(ndcg_at_k(recs, rel, k))
            per[k]["recall"].append(recall_at_k(recs, rel, k))
        mrrs.append(mrr(recs, rel))
    for k in k_values:
        for name, v in per[k].items():
            res[f"{name}@{k}"] = float(sum(v) / len(v)) if v else 0.0
    res["mrr"] = float(sum(mrrs) / len(mrrs)) if mrrs else 0.0
    return res
