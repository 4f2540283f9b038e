"""Restatement of the ranking-feature glue between retrieval and ranking (TEST INFRASTRUCTURE).

Follows /root/reference/src/serving/recommender.py:213-263 (_build_ranking_features; duplicated
inline at src/pipelines/run_pipeline.py:188-213) and the 50-column order of
src/features/feature_engineering.py:434-443 (get_feature_columns).  Python floats (float64) exactly as
the reference computes them; the cast to float32 happens in LightGBMRanker.predict (ranker.py:173).
Pinned by tests/golden/g8_ranking_features.npz generated from the reference's own function.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np

N_GENRES = 18
USER_SCALARS = [("avg_rating", 3.5), ("log_rating_count", 0.0), ("recency_score", 0.5), ("gender_encoded", 0.0),
                ("age_normalized", 0.3), ("occupation_normalized", 0.3)]
ITEM_SCALARS = [("avg_rating", 3.5), ("log_rating_count", 0.0), ("popularity_score", 0.0), ("rating_stddev", 0.0),
                ("year_normalized", 0.5)]
ITEM_COLS = ["item_avg_rating", "item_log_rating_count", "popularity_score", "rating_stddev", "year_normalized"]


def feature_columns() -> List[str]:
    """feature_engineering.py:434-443"""
    return (["avg_rating", "log_rating_count", "recency_score", "gender_encoded", "age_normalized",
             "occupation_normalized", "item_avg_rating", "item_log_rating_count", "popularity_score", "rating_stddev",
             "year_normalized", "rating_diff", "user_item_popularity_ratio", "genre_affinity"]
            + [f"user_genre_{i}" for i in range(N_GENRES)] + [f"item_genre_{i}" for i in range(N_GENRES)])


def build_ranking_features(user_features: Dict[str, Any], item_features_batch: Dict[int, Optional[Dict[str, Any]]],
                           candidate_item_ids: List[int]) -> Dict[str, np.ndarray]:
    """Returns {column -> float64[n]} with exactly the reference's columns (item_id included)."""
    rows = []
    for item_id in candidate_item_ids:
        item_feat = item_features_batch.get(item_id) or {}
        row = {"item_id": item_id}
        for name, dflt in USER_SCALARS:
            row[name] = float(user_features.get(name, dflt))
        for (name, dflt), col in zip(ITEM_SCALARS, ITEM_COLS):
            row[col] = float(item_feat.get(name, dflt))
        row["rating_diff"] = row["avg_rating"] - row["item_avg_rating"]
        row["user_item_popularity_ratio"] = row["log_rating_count"] / (row["item_log_rating_count"] + 1e-8)
        ug = user_features.get("genre_pref", [0.0] * N_GENRES)
        ig = item_feat.get("genre_vector", [0.0] * N_GENRES)
        for i in range(N_GENRES):
            row[f"user_genre_{i}"] = float(ug[i]) if i < len(ug) else 0.0
            row[f"item_genre_{i}"] = float(ig[i]) if i < len(ig) else 0.0
        row["genre_affinity"] = sum(row[f"user_genre_{i}"] * row[f"item_genre_{i}"] for i in range(N_GENRES))
        rows.append(row)
    cols = list(rows[0].keys()) if rows else []
    return {c: np.array([r[c] for r in rows], dtype=np.float64) for c in cols}


def feature_matrix(cols: Dict[str, np.ndarray], feature_names: List[str]) -> np.ndarray:
    """What ranker.predict feeds the forest: df[feature_names] (missing columns -> 0.0, recommender.py:334-336)
    cast to float32 (ranker.py:173)."""
    n = len(next(iter(cols.values()))) if cols else 0
    X = np.zeros((n, len(feature_names)), dtype=np.float64)
    for j, name in enumerate(feature_names):
        if name in cols:
            X[:, j] = cols[name]
    return X.astype(np.float32)
