"""Seeded synthetic data of the shapes BASELINE.json names (there is no MovieLens offline).

ml1m_like(): a MovieLens-1M-*shaped* ratings table (6 040 users, 3 952-wide item-id space with
3 883 catalogue items, ~1.0 M ratings, ML-1M rating histogram, per-user increasing timestamps,
18-genre multi-hot) drawn from a low-rank latent model so that ranking quality is learnable.
File formats follow the reference loader (src/features/feature_engineering.py:43-65)."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Tuple

import numpy as np
import pandas as pd

GENRES = ["Action", "Adventure", "Animation", "Children's", "Comedy", "Crime", "Documentary", "Drama", "Fantasy",
          "Film-Noir", "Horror", "Musical", "Mystery", "Romance", "Sci-Fi", "Thriller", "War", "Western"]
N_GENRES = len(GENRES)


def ml1m_like(n_users: int = 6040, n_item_ids: int = 3952, n_catalog: int = 3883, n_ratings: int = 1_000_209,
              rank: int = 8, seed: int = 0, affinity: float = 0.8, shuffle_time: bool = False) -> Tuple[pd.DataFrame, pd.DataFrame, np.ndarray]:
    """Returns (ratings_df[user_id,item_id,rating,timestamp], movies_df[item_id,title,genres], genre_matrix
    [n_item_ids+1,18] f32).  Deterministic for a given seed.
    `affinity` weights the latent user-item affinity against popularity when a user picks what to rate.  Ratings are
    time-stamped in draw order by default (weighted sampling without replacement draws a user's strongest items
    first, so the "last 10 % by timestamp" test split of the reference's run_evaluate holds the user's weakest picks
    and NDCG@10 stays ~0.02 -- the G9 fixture was generated on exactly this set); `shuffle_time=True` stamps them in
    a random order instead.  Both were tried for G9: with the reference's 10 epochs at lr 1e-3 the model mostly learns
    popularity either way (NDCG@10 0.016-0.02), so the per-epoch loss curve remains the tight parity check."""
    rng = np.random.RandomState(seed)
    catalog = np.sort(rng.choice(np.arange(1, n_item_ids + 1), size=n_catalog, replace=False))
    zu = rng.randn(n_users + 1, rank).astype(np.float32)
    zi = rng.randn(n_item_ids + 1, rank).astype(np.float32)
    pop = rng.zipf(1.3, size=n_catalog).astype(np.float64)
    pop = pop / pop.sum()
    # per-user activity ~ log-normal, at least 20 ratings like ML-1M
    act = np.maximum(20, rng.lognormal(mean=4.6, sigma=0.9, size=n_users)).astype(np.float64)
    act = np.minimum(act, n_catalog * 0.6)
    act = act * (n_ratings / act.sum())
    cnt = np.maximum(20, np.round(act).astype(np.int64))
    users, items = [], []
    for u in range(1, n_users + 1):
        c = int(min(cnt[u - 1], n_catalog))
        # preference-biased choice without replacement: popularity x exp(affinity)
        aff = zi[catalog] @ zu[u]
        w = pop * np.exp(affinity * (aff - aff.max()))
        w /= w.sum()
        it = rng.choice(catalog, size=c, replace=False, p=w)
        if shuffle_time:
            it = rng.permutation(it)
        users.append(np.full(c, u, dtype=np.int64))
        items.append(it.astype(np.int64))
    users = np.concatenate(users)
    items = np.concatenate(items)
    score = (zu[users] * zi[items]).sum(1) / np.sqrt(rank) + 0.6 * rng.randn(users.size)
    # ML-1M histogram {1:5.6%, 2:10.8%, 3:26.1%, 4:34.9%, 5:22.6%} through score quantiles
    qs = np.quantile(score, [0.056, 0.164, 0.425, 0.774])
    rating = 1 + (score[:, None] > qs[None, :]).sum(1)
    # timestamps: increasing within each user in generation order
    order = np.arange(users.size)
    ts = 956_703_932 + order * 7
    ratings = pd.DataFrame({"user_id": users, "item_id": items, "rating": rating.astype(np.int64), "timestamp": ts})
    gm = np.zeros((n_item_ids + 1, N_GENRES), dtype=np.float32)
    titles, gstr = [], []
    for it in catalog:
        k = rng.randint(1, 4)
        gs = np.sort(rng.choice(N_GENRES, size=k, replace=False))
        gm[it, gs] = 1.0
        titles.append(f"Movie {it} ({1930 + int(it) % 70})")
        gstr.append("|".join(GENRES[g] for g in gs))
    movies = pd.DataFrame({"item_id": catalog.astype(np.int64), "title": titles, "genres": gstr})
    return ratings, movies, gm


def write_ml1m_files(out_dir: str, ratings: pd.DataFrame, movies: pd.DataFrame, n_users: int) -> None:
    """ratings.dat / movies.dat / users.dat in the '::' format the reference loader reads."""
    p = Path(out_dir)
    p.mkdir(parents=True, exist_ok=True)
    with open(p / "ratings.dat", "w", encoding="latin-1") as f:
        for u, i, r, t in ratings[["user_id", "item_id", "rating", "timestamp"]].itertuples(index=False):
            f.write(f"{u}::{i}::{r}::{t}\n")
    with open(p / "movies.dat", "w", encoding="latin-1") as f:
        for i, t, g in movies[["item_id", "title", "genres"]].itertuples(index=False):
            f.write(f"{i}::{t}::{g}\n")
    with open(p / "users.dat", "w", encoding="latin-1") as f:
        for u in range(1, n_users + 1):
            f.write(f"{u}::{'MF'[u % 2]}::{[1, 18, 25, 35, 45, 50, 56][u % 7]}::{u % 21}::{10000 + u % 89999}\n")


# ---- synthetic LambdaMART forest in LightGBM's text format (bench / tests data generator) ----------------------
def write_text_model(model: Dict) -> str:
    """Emit a LightGBM-format text model (the format LightGBMRanker.load reads; reference ranker.py:209,:219)."""
    names = model["feature_names"]
    hdr = ["tree", "version=v4", "num_class=1", "num_tree_per_iteration=1", "label_index=0",
           f"max_feature_idx={len(names) - 1}", "objective=lambdarank",
           "feature_names=" + " ".join(names),
           "feature_infos=" + " ".join(["[-1e30:1e30]"] * len(names)), "tree_sizes=0", ""]
    body = []
    for i, t in enumerate(model["trees"]):
        body.append(f"Tree={i}")
        body.append(f"num_leaves={t['num_leaves']}")
        body.append(f"num_cat={t.get('num_cat', 0)}")
        if t["num_leaves"] > 1:
            body.append("split_feature=" + " ".join(str(int(x)) for x in t["split_feature"]))
            body.append("split_gain=" + " ".join("1" for _ in t["split_feature"]))
            body.append("threshold=" + " ".join(repr(float(x)) for x in t["threshold"]))
            body.append("decision_type=" + " ".join(str(int(x)) for x in t["decision_type"]))
            body.append("left_child=" + " ".join(str(int(x)) for x in t["left_child"]))
            body.append("right_child=" + " ".join(str(int(x)) for x in t["right_child"]))
        body.append("leaf_value=" + " ".join(repr(float(x)) for x in t["leaf_value"]))
        if t["num_leaves"] > 1:
            if t.get("num_cat", 0) > 0:
                body.append("cat_boundaries=" + " ".join(str(int(x)) for x in t["cat_boundaries"]))
                body.append("cat_threshold=" + " ".join(str(int(x)) for x in t["cat_threshold"]))
        body.append("is_linear=0")
        body.append(f"shrinkage={t.get('shrinkage', 1.0)}")
        body.append("")
        body.append("")
    tail = ["end of trees", "", "feature_importances:", "", "parameters:", "[boosting: gbdt]",
            "end of parameters", "", "pandas_categorical:null", ""]
    return "\n".join(hdr + body + tail)


def random_forest_model(n_trees: int, n_leaves: int, n_features: int, seed: int = 4, names=None) -> Dict:
    """Synthetic forest: random (unbalanced) binary trees grown leaf-by-leaf like
    LightGBM's best-first growth; thresholds ~ N(0,1) quantiles; decision_type=2."""
    rng = np.random.RandomState(seed)
    names = names or [f"Column_{i}" for i in range(n_features)]
    trees = []
    for _ in range(n_trees):
        nl = n_leaves
        sf = np.zeros(nl - 1, np.int64)
        th = np.zeros(nl - 1, np.float64)
        lc = np.zeros(nl - 1, np.int64)
        rc = np.zeros(nl - 1, np.int64)
        # start: node 0 with leaves 0 (left) / 1 (right)
        lc[0], rc[0] = ~0, ~1
        sf[0] = rng.randint(n_features)
        th[0] = rng.randn()
        # parent pointers to patch when a leaf is split
        leaf_parent = {0: (0, 0), 1: (0, 1)}  # leaf -> (node, side)
        for new_node in range(1, nl - 1):
            leaf = rng.choice(list(leaf_parent.keys()))
            pn, side = leaf_parent.pop(leaf)
            if side == 0:
                lc[pn] = new_node
            else:
                rc[pn] = new_node
            new_leaf = new_node + 1
            lc[new_node], rc[new_node] = ~leaf, ~new_leaf
            leaf_parent[leaf] = (new_node, 0)
            leaf_parent[new_leaf] = (new_node, 1)
            sf[new_node] = rng.randint(n_features)
            th[new_node] = rng.randn()
        trees.append(dict(num_leaves=nl, num_cat=0, split_feature=sf, threshold=th,
                          decision_type=np.full(nl - 1, 2, np.int64), left_child=lc, right_child=rc,
                          leaf_value=rng.randn(nl) * 0.05, shrinkage=0.05))
    return dict(feature_names=names, max_feature_idx=n_features - 1, num_class=1,
                num_tree_per_iteration=1, average_output=False, objective="lambdarank", trees=trees)
