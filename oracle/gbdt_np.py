"""NumPy restatement of the LambdaMART forward (TEST INFRASTRUCTURE).

Follows the call sites in /root/reference/src/models/ranker.py (:161-174 predict
-> Booster.predict raw score; :212-226 load -> Booster(model_file=...)).  The
arithmetic lives in lightgbm (>=4.1.0, requirements.txt:3), third-party and not
installed here: the text-model format and decision rule are restated from
LightGBM's published behaviour (Tree::NumericalDecision / CategoricalDecision,
GBDT::SaveModelToString).  **Parity unpinned**: pinned only by a hand-written
model with hand-computed scores (tests/golden/tiny_forest.txt).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

K_ZERO = 1e-35  # LightGBM kZeroThreshold


def parse_text_model(text: str) -> Dict:
    lines = text.splitlines()
    header: Dict[str, str] = {}
    trees: List[Dict] = []
    cur = None
    in_trees = False
    for ln in lines:
        ln = ln.strip()
        if ln.startswith("Tree="):
            cur = {}
            trees.append(cur)
            in_trees = True
            continue
        if ln == "end of trees":
            cur = None
            break
        if "=" not in ln:
            continue
        k, v = ln.split("=", 1)
        if in_trees and cur is not None:
            cur[k] = v
        else:
            header[k] = v

    def ints(s):
        return np.array([int(x) for x in s.split()], dtype=np.int64) if s else np.zeros(0, np.int64)

    def flts(s):
        return np.array([float(x) for x in s.split()], dtype=np.float64) if s else np.zeros(0, np.float64)

    out_trees = []
    for t in trees:
        nl = int(t["num_leaves"])
        d = dict(num_leaves=nl, num_cat=int(t.get("num_cat", "0")),
                 leaf_value=flts(t.get("leaf_value", "")),
                 shrinkage=float(t.get("shrinkage", "1")))
        if nl > 1:
            d.update(split_feature=ints(t["split_feature"]), threshold=flts(t["threshold"]),
                     decision_type=ints(t["decision_type"]), left_child=ints(t["left_child"]),
                     right_child=ints(t["right_child"]))
            if d["num_cat"] > 0:
                d.update(cat_boundaries=ints(t["cat_boundaries"]), cat_threshold=ints(t["cat_threshold"]))
        out_trees.append(d)
    names = header.get("feature_names", "").split()
    return dict(feature_names=names, max_feature_idx=int(header.get("max_feature_idx", len(names) - 1)),
                num_class=int(header.get("num_class", "1")),
                num_tree_per_iteration=int(header.get("num_tree_per_iteration", "1")),
                average_output=("average_output" in text.split("Tree=")[0]),
                objective=header.get("objective", ""), trees=out_trees)


def _decide_left(fval: np.ndarray, t: Dict, node: int) -> np.ndarray:
    dt = int(t["decision_type"][node])
    thr = t["threshold"][node]
    if dt & 1:  # categorical
        left = np.zeros(fval.shape, dtype=bool)
        ok = ~np.isnan(fval)
        iv = np.where(ok, fval, -1).astype(np.int64)
        ok &= iv >= 0
        ci = int(thr)
        b0, b1 = int(t["cat_boundaries"][ci]), int(t["cat_boundaries"][ci + 1])
        words = t["cat_threshold"][b0:b1]
        w = iv // 32
        inb = ok & (w < (b1 - b0))
        bits = np.zeros(fval.shape, dtype=np.int64)
        bits[inb] = (words[w[inb]] >> (iv[inb] % 32)) & 1
        left[inb] = bits[inb] == 1
        return left
    default_left = bool(dt & 2)
    missing = (dt >> 2) & 3  # 0 none, 1 zero, 2 nan
    f = fval.copy()
    isn = np.isnan(f)
    if missing != 2:
        f = np.where(isn, 0.0, f)
    if missing == 1:
        is_missing = np.abs(f) <= K_ZERO
    elif missing == 2:
        is_missing = isn
    else:
        is_missing = np.zeros(f.shape, dtype=bool)
    with np.errstate(invalid="ignore"):
        le = f <= thr
    return np.where(is_missing, default_left, le)


def predict_raw(model: Dict, X: np.ndarray) -> np.ndarray:
    """Sum over trees of the reached leaf value -> float64[n] (Booster.predict raw)."""
    X = np.asarray(X, dtype=np.float32).astype(np.float64)  # ranker.py:173 casts to f32 first
    n = X.shape[0]
    out = np.zeros(n, dtype=np.float64)
    for t in model["trees"]:
        if t["num_leaves"] <= 1:
            out += t["leaf_value"][0] if t["leaf_value"].size else 0.0
            continue
        node = np.zeros(n, dtype=np.int64)
        active = np.ones(n, dtype=bool)
        while active.any():
            for nd in np.unique(node[active]):
                sel = active & (node == nd)
                left = _decide_left(X[sel, t["split_feature"][nd]], t, int(nd))
                node[sel] = np.where(left, t["left_child"][nd], t["right_child"][nd])
            active = node >= 0
        out += t["leaf_value"][~node]
    if model.get("average_output") and model["trees"]:
        out /= len(model["trees"])
    return out


def write_text_model(model: Dict) -> str:
    """Emit a LightGBM-format text model (used to synthesise bench/test forests)."""
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
