"""Restatement of the parity metric (TEST INFRASTRUCTURE).

Follows /root/reference/src/evaluation/metrics.py:20-69 (ndcg_at_k, binary
relevance), :72-100 (recall_at_k), mrr, coverage, and :301-384 evaluate_model
(mean over users with >=1 relevant item).  Pinned by the reference's
known-answer tests (tests/test_models.py:372-426) in tests/test_oracle.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence


def ndcg_at_k(recommended: Sequence, relevant: Sequence, k: int) -> float:
    rel = set(relevant)
    dcg = sum(1.0 / math.log2(i + 2) for i, it in enumerate(list(recommended)[:k]) if it in rel)
    idcg = sum(1.0 / math.log2(i + 2) for i in range(min(len(relevant), k)))
    return 0.0 if idcg == 0 else dcg / idcg


def recall_at_k(recommended: Sequence, relevant: Sequence, k: int) -> float:
    if not relevant:
        return 0.0
    rel = set(relevant)
    return sum(1 for it in list(recommended)[:k] if it in rel) / len(rel)


def mrr(recommended: Sequence, relevant: Sequence) -> float:
    rel = set(relevant)
    for i, it in enumerate(recommended):
        if it in rel:
            return 1.0 / (i + 1)
    return 0.0


def coverage(recs: List[Sequence], catalog_size: int) -> float:
    seen = set()
    for r in recs:
        seen.update(r)
    return len(seen) / catalog_size if catalog_size else 0.0


def mean_ndcg(recs: Dict[int, Sequence], truth: Dict[int, Sequence], k: int) -> float:
    vals = [ndcg_at_k(recs[u], truth[u], k) for u in recs if truth.get(u)]
    return sum(vals) / len(vals) if vals else 0.0
