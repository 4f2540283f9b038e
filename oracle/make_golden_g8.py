#!/usr/bin/env python3
"""G8: outputs of the reference's own RecommendationPipeline._build_ranking_features for fixed feature dicts
(imported in this container only; `pydantic_settings` is replaced by an empty in-process stand-in so that
src.config imports) -> tests/golden/g8_ranking_features.npz.  Usage: python oracle/make_golden_g8.py"""
import json
import sys
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import pydantic  # noqa: E402

stub = types.ModuleType("pydantic_settings")


class BaseSettings(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="ignore")


stub.BaseSettings = BaseSettings
stub.SettingsConfigDict = dict
sys.modules["pydantic_settings"] = stub
sys.path.insert(0, "/root/reference")


def make_inputs(seed, n_items=40, n_cand=25):
    rng = np.random.RandomState(seed)
    user = {"avg_rating": float(rng.uniform(1, 5)), "log_rating_count": float(rng.uniform(0, 8)),
            "recency_score": float(rng.rand()), "gender_encoded": float(rng.randint(2)),
            "age_normalized": float(rng.rand()), "occupation_normalized": float(rng.rand()),
            "genre_pref": [float(x) for x in rng.rand(18)]}
    if seed % 3 == 1:   # sparse user record -> defaults
        user = {"avg_rating": user["avg_rating"]}
    items = {}
    for i in range(1, n_items + 1):
        if rng.rand() < 0.2:
            items[i] = None          # missing from the store -> all defaults
            continue
        items[i] = {"avg_rating": float(rng.uniform(1, 5)), "log_rating_count": float(rng.uniform(0, 9)),
                    "popularity_score": float(rng.rand()), "rating_stddev": float(rng.rand() * 1.5),
                    "year_normalized": float(rng.rand()),
                    "genre_vector": [float(x) for x in (rng.rand(18) < 0.15)]}
        if rng.rand() < 0.15:
            items[i]["log_rating_count"] = 0.0   # exercises the 1e-8 guard of the ratio
    cand = [int(x) for x in rng.choice(np.arange(1, n_items + 6), size=n_cand, replace=False)]  # some ids unknown
    return user, items, cand


def main():
    from src.serving.recommender import RecommendationPipeline
    out = {}
    meta = []
    for seed in range(4):
        user, items, cand = make_inputs(seed)
        df = RecommendationPipeline._build_ranking_features(None, user, items, cand)
        out[f"s{seed}_columns"] = np.array(list(df.columns))
        out[f"s{seed}_values"] = df.values.astype(np.float64)
        meta.append({"seed": seed, "user": user, "items": {str(k): v for k, v in items.items()}, "cand": cand})
    np.savez_compressed(ROOT / "tests" / "golden" / "g8_ranking_features.npz", **out)
    (ROOT / "tests" / "golden" / "g8_inputs.json").write_text(json.dumps(meta))
    print("wrote g8:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
