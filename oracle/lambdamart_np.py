"""NumPy restatement of the LambdaMART trainer (TEST INFRASTRUCTURE).

What the reference trains: ``lgb.train(params, ...)`` with ``objective=lambdarank`` (src/models/ranker.py:86-151:
num_leaves 63, learning_rate 0.05, min_child_samples 20, colsample_bytree 0.8, reg_alpha 0.1, reg_lambda 0.1,
label_gain [0,1,3,7,15], eval_at [5,10,20], early stopping 30 rounds on the validation NDCG; ``subsample`` 0.8 is
inactive in LightGBM without ``bagging_freq``).  The arithmetic lives in lightgbm (>=4.1.0, requirements.txt:3), which is
third-party and not installed here: **parity unpinned**.  This file restates LightGBM's published algorithm --
histogram-based leaf-wise GBDT (``SerialTreeLearner``: per-leaf (gradient, hessian, count) histograms over <= 255
bins per feature, histogram subtraction, gain = ThresholdL1(G)^2/(H+l2), best-first growth) with the lambdarank
objective (``LambdarankNDCG::GetGradientsForOneQuery``: pairs (i < truncation_level, j > i) with different labels,
delta NDCG x sigmoid, normalisation by (0.01 + |delta score|) and by log2(1+sum)/sum) -- in the exact form the HIP
trainer (recommendit_amd/csrc/gbdt_train.hip) implements it, so that the two can be compared tree by tree:
  * bin upper bounds from a strided sample of <= 200 000 rows (LightGBM: bin_construct_sample_cnt), midpoints between
    neighbouring distinct values, equal-frequency cuts when a feature has more than max_bin distinct values;
  * gradients / hessians are quantised to integers (2^20 levels of the largest magnitude) before they enter the
    histograms, so every histogram sum is an integer and independent of the summation order: splits are bit-exact;
  * feature_fraction picks the features of a tree with a counter-based generator (splitmix64 of seed, tree, feature).

Fidelity branches (params ``hist_dtype``, ``use_missing``, ``split_order``; the defaults are the round-2 behaviour above).
These are written from LightGBM's published ``FeatureHistogram::FindBestThreshold`` / ``FindBestThresholdSequentially``,
NOT from the kernel (which evaluates every threshold in parallel from prefix sums):
  * ``hist_dtype="float"``: no quantisation -- gradients / hessians enter the histograms as they are and every sum is an
    f64 accumulation in row order (LightGBM: float gradients, double histogram entries).  The HIP trainer's
    ``hist_bits=40`` mode (2^-40 fixed point, order-independent) is compared against THIS branch with a tolerance;
    ``hist_dtype="int40"`` restates the fixed-point mode itself bit for bit;
  * ``use_missing=True``: a feature whose bin sample holds a NaN gets a last "missing" bin (missing type NaN); the
    threshold search runs LightGBM's two sequential scans -- right-to-left accumulating the RIGHT side (the missing rows
    fall to the left: default_left) and left-to-right accumulating the LEFT side (missing rows right), the second scan
    also offering "every real value left | missing right" (its mirror image "missing rows alone left | every real value
    right" has the same gain up to rounding and is not offered: which of the two a float comparison prefers is noise);
  * ``split_order="lightgbm"``: a candidate replaces the incumbent only if its gain is strictly larger, in LightGBM's
    scan order (right-to-left first), so equal gains -- runs of empty bins -- keep the HIGHEST threshold; ``"low"`` keeps
    the lowest.  Across features the lower feature index wins a tie in both (SplitInfo::operator>).
Still different from LightGBM (parity unpinned: lightgbm is not importable here): its bin finder (GreedyFindBin /
zero-as-one-bin / min_data_in_bin), per-bin counts estimated from hessians, and its own feature_fraction RNG.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

K_EPS = 1e-15
QLEVELS = float(1 << 20)
MASK64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & MASK64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & MASK64
    x ^= x >> 31
    return x


def default_params(**kw) -> Dict:
    p = dict(num_leaves=63, n_estimators=500, learning_rate=0.05, min_child_samples=20, max_bin=255,
             truncation_level=30, early_stopping_rounds=30, eval_at=[5, 10, 20], reg_alpha=0.1, reg_lambda=0.1,
             feature_fraction=0.8, min_sum_hessian=1e-3, sigmoid=1.0, label_gain=[0, 1, 3, 7, 15], seed=2,
             lambdarank_norm=True, bin_sample=200000, hist_dtype="int20", use_missing=False, split_order="low")
    p.update(kw)
    return p


# ------------------------------------------------------------------ binning
def find_bin_bounds(X: np.ndarray, max_bin: int, bin_sample: int, use_missing: bool = False):
    """-> (bounds per feature, nanbin per feature): nanbin[f] = index of the feature's "missing" bin (the last one) or -1"""
    n = X.shape[0]
    step = max(1, (n + bin_sample - 1) // bin_sample)
    S = X[::step]
    out, nanbin = [], []
    for f in range(X.shape[1]):
        v = np.sort(S[:, f].astype(np.float64))
        has_nan = bool(use_missing and np.isnan(v).any())     # LightGBM: missing type NaN iff the bin sample holds one
        max_real = max_bin - 1 if has_nan else max_bin
        v = v[~np.isnan(v)]
        u, c = np.unique(v, return_counts=True)
        if u.size <= 1:
            ub = np.array([])
        elif u.size <= max_real:
            ub = (u[:-1] + u[1:]) * 0.5
        else:   # equal-frequency cuts on the cumulative counts, each cut placed between two distinct values
            cum = np.cumsum(c)
            tot = cum[-1]
            cuts, last = [], -1
            for b in range(1, max_real):
                i = int(np.searchsorted(cum, (tot * b + max_real - 1) // max_real, side="left"))
                i = min(i, u.size - 2)
                if i > last:
                    cuts.append(i)
                    last = i
            ub = np.array([(u[i] + u[i + 1]) * 0.5 for i in cuts])
        b = np.concatenate([ub, [np.inf]])
        nanbin.append(len(b) if has_nan else -1)
        if has_nan:
            b = np.concatenate([b, [np.nan]])
        out.append(b)
    return out, nanbin


def bin_matrix(X: np.ndarray, bounds: List[np.ndarray], nanbin: Optional[Sequence[int]] = None) -> np.ndarray:
    B = np.zeros(X.shape, dtype=np.uint8)
    for f, ub in enumerate(bounds):
        x = X[:, f].astype(np.float64)
        nb_ = -1 if nanbin is None else nanbin[f]
        isn = np.isnan(x)
        real = ub[:nb_] if nb_ >= 0 else ub
        b = np.searchsorted(real, np.where(isn, 0.0, x), side="left")   # first bin with x <= upper bound
        B[:, f] = np.where(isn & (nb_ >= 0), nb_, b).astype(np.uint8)
    return B


# ------------------------------------------------------------------ lambdarank objective
def _max_dcg(labels: np.ndarray, k: int, gain: np.ndarray) -> float:
    g = np.sort(gain[labels.astype(np.int64)])[::-1][:k]
    return float((g / np.log2(np.arange(g.size) + 2.0)).sum())


def lambdarank_grads(scores: np.ndarray, labels: np.ndarray, groups: Sequence[int], p: Dict) -> Tuple[np.ndarray, np.ndarray]:
    gain = np.asarray(p["label_gain"], dtype=np.float64)
    sig, T = float(p["sigmoid"]), int(p["truncation_level"])
    lam = np.zeros(scores.shape, dtype=np.float64)
    hes = np.zeros(scores.shape, dtype=np.float64)
    b = 0
    for cnt in groups:
        s = scores[b:b + cnt].astype(np.float64)
        lab = labels[b:b + cnt].astype(np.int64)
        mx = _max_dcg(lab, T, gain)
        inv = 1.0 / mx if mx > 0 else 0.0
        order = np.argsort(-s, kind="stable")                 # ties keep the original order
        ss, ll = s[order], lab[order]
        disc = 1.0 / np.log2(np.arange(cnt) + 2.0)
        gl = gain[ll]
        l_sorted = np.zeros(cnt)
        h_sorted = np.zeros(cnt)
        norm = bool(p["lambdarank_norm"])
        best, worst = ss[0], ss[-1]
        sum_l = 0.0
        for i in range(min(cnt - 1, T)):
            j = np.arange(i + 1, cnt)
            m = ll[j] != ll[i]
            if not m.any():
                continue
            j = j[m]
            hi_is_i = ll[i] > ll[j]
            ds = np.where(hi_is_i, ss[i] - ss[j], ss[j] - ss[i])          # high score - low score
            dn = np.abs(gl[i] - gl[j]) * np.abs(disc[i] - disc[j]) * inv
            if norm and best != worst:
                dn = dn / (0.01 + np.abs(ds))
            rho = 1.0 / (1.0 + np.exp(sig * ds))
            pl = -sig * dn * rho                                           # p_lambda (<= 0)
            ph = sig * sig * dn * rho * (1.0 - rho)
            # high gets +p_lambda, low gets -p_lambda
            l_sorted[i] += np.where(hi_is_i, pl, -pl).sum()
            np.add.at(l_sorted, j, np.where(hi_is_i, -pl, pl))
            h_sorted[i] += ph.sum()
            np.add.at(h_sorted, j, ph)
            sum_l += float((-2.0 * pl).sum())
        if norm and sum_l > 0:
            nf = np.log2(1.0 + sum_l) / sum_l
            l_sorted *= nf
            h_sorted *= nf
        lam[b + order] = l_sorted
        hes[b + order] = h_sorted
        b += cnt
    return lam, hes


def quantize(g: np.ndarray, h: np.ndarray, levels: float = QLEVELS) -> Tuple[np.ndarray, np.ndarray, float, float]:
    gm, hm = float(np.abs(g).max()), float(h.max())
    sg = levels / gm if gm > 0 else 0.0
    sh = levels / hm if hm > 0 else 0.0
    return np.rint(g * sg).astype(np.int64), np.rint(h * sh).astype(np.int64), sg, sh


def ndcg_at(scores: np.ndarray, labels: np.ndarray, groups: Sequence[int], ks: Sequence[int], gain: np.ndarray) -> List[float]:
    tot = np.zeros(len(ks))
    b = 0
    for cnt in groups:
        s, lab = scores[b:b + cnt], labels[b:b + cnt].astype(np.int64)
        order = np.argsort(-s.astype(np.float64), kind="stable")
        g = gain[lab[order]]
        for t, k in enumerate(ks):
            mx = _max_dcg(lab, k, gain)
            if mx <= 0:
                tot[t] += 1.0
            else:
                kk = min(k, cnt)
                tot[t] += float((g[:kk] / np.log2(np.arange(kk) + 2.0)).sum()) / mx
        b += cnt
    return (tot / len(groups)).tolist()


# ------------------------------------------------------------------ tree learner
def _thr_l1(g: float, l1: float) -> float:
    return np.sign(g) * max(abs(g) - l1, 0.0)


def _leaf_gain(G: float, H: float, l1: float, l2: float) -> float:
    t = _thr_l1(G, l1)
    return t * t / (H + l2)


def _best_split(hist: np.ndarray, nb: Sequence[int], used: np.ndarray, sg: float, sh: float, p: Dict,
                nanbin: Optional[Sequence[int]] = None):
    """hist [F, 256, 3] (g, h, count; integers scaled by sg / sh, or raw f64 sums with sg = sh = 1)
    -> (gain, feature, bin, GL, HL, CL, default_left) or None.  Left = real bins <= bin (+ the missing bin if default_left).

    Two sequential scans per feature, as FeatureHistogram::FindBestThresholdSequentially runs them: right-to-left
    accumulating the right child (left = total - right, so the missing bin lands on the left), then -- only for a feature
    with a missing bin -- left-to-right accumulating the left child (missing rows right).  A candidate replaces the
    incumbent only on a strictly larger gain; ``split_order`` decides the order of evaluation (see the module doc)."""
    l1, l2 = float(p["reg_alpha"]), float(p["reg_lambda"])
    lgb_order = p.get("split_order", "low") == "lightgbm"
    flt = hist.dtype.kind == "f"
    num = (lambda v: float(v)) if flt else (lambda v: int(v))
    best = None
    for f in range(hist.shape[0]):
        if not used[f] or nb[f] < 2:
            continue
        nbn = -1 if nanbin is None else nanbin[f]
        nr = nb[f] - 1 if nbn >= 0 else nb[f]                     # real bins
        tq = hist[f, :nb[f]].sum(0)
        TG, TH, TC = num(tq[0]), num(tq[1]), int(tq[2])
        G, H = (TG / sg if sg else 0.0), (TH / sh if sh else 0.0)
        parent = _leaf_gain(G, H, l1, l2)
        cand = []                                                  # (lg, lh, lc, bin, default_left) in evaluation order
        # right-to-left: threshold t-1 for t = nr-1 .. 1; right = real bins >= t
        rg = rh = 0 if not flt else 0.0
        rc = 0
        rev = []
        for t in range(nr - 1, 0, -1):
            rg += num(hist[f, t, 0]); rh += num(hist[f, t, 1]); rc += int(hist[f, t, 2])
            if nbn >= 0 and (TC - rc) - int(hist[f, nbn, 2]) == 0:
                continue    # the missing rows alone on the left: mirror image of the last left-to-right candidate, offered there
            rev.append((TG - rg, TH - rh, TC - rc, t - 1, True))
        # left-to-right (missing bin only): threshold t for t = 0 .. nr-1; left = real bins <= t
        fwd = []
        if nbn >= 0:
            lg = lh = 0 if not flt else 0.0
            lc = 0
            for t in range(nr):
                lg += num(hist[f, t, 0]); lh += num(hist[f, t, 1]); lc += int(hist[f, t, 2])
                fwd.append((lg, lh, lc, t, False))
        if lgb_order:
            cand = rev + fwd
        elif nbn < 0:
            cand = rev[::-1]                                       # lowest threshold first
        else:                                                      # lowest threshold first, missing-left before missing-right
            by_bin = {}
            for c in rev[::-1]:
                by_bin.setdefault(c[3], []).append(c)
            for c in fwd:
                by_bin.setdefault(c[3], []).append(c)
            cand = [c for b in sorted(by_bin) for c in by_bin[b]]
        fbest = None
        for (lg, lh, lc, b, dl) in cand:
            GL, HL = (lg / sg if sg else 0.0), (lh / sh if sh else 0.0)
            GR, HR, CR = G - GL, H - HL, TC - lc
            if lc < p["min_child_samples"] or CR < p["min_child_samples"]:
                continue
            if HL < p["min_sum_hessian"] or HR < p["min_sum_hessian"]:
                continue
            gain = _leaf_gain(GL, HL, l1, l2) + _leaf_gain(GR, HR, l1, l2) - parent
            if gain > K_EPS and (fbest is None or gain > fbest[0]):
                fbest = (gain, f, b, lg, lh, lc, dl)
        if fbest is not None and (best is None or fbest[0] > best[0]):    # ties between features: the lower index
            best = fbest
    return best


def train(X: np.ndarray, y: np.ndarray, groups: Sequence[int], params: Optional[Dict] = None,
          Xv: Optional[np.ndarray] = None, yv: Optional[np.ndarray] = None, groups_v: Optional[Sequence[int]] = None,
          feature_names: Optional[List[str]] = None) -> Dict:
    p = default_params(**(params or {}))
    n, F = X.shape
    gain_tab = np.asarray(p["label_gain"], dtype=np.float64)
    bounds, nanbin = find_bin_bounds(X, p["max_bin"], p["bin_sample"], bool(p["use_missing"]))
    nb = [len(b) for b in bounds]
    Xb = bin_matrix(X, bounds, nanbin)
    Xvb = bin_matrix(Xv, bounds, nanbin) if Xv is not None else None
    mode = p["hist_dtype"]
    assert mode in ("int20", "int40", "float")
    levels = QLEVELS
    if mode == "int40":   # 2^40 levels while n of them stay below 2^62
        levels = float(1 << min(40, 62 - int(np.ceil(np.log2(max(n, 2))))))
    scores = np.zeros(n)
    vscores = np.zeros(len(Xv)) if Xv is not None else None
    trees, history = [], []
    best_val, best_it, best_itr = None, 0, None
    n_used = max(1, int(F * p["feature_fraction"] + 0.5)) if p["feature_fraction"] < 1.0 else F
    for it in range(p["n_estimators"]):
        lam, hes = lambdarank_grads(scores, y, groups, p)
        if mode == "float":     # LightGBM's default: unquantised gradients, f64 histogram sums
            gq, hq, sg, sh = lam.astype(np.float64), hes.astype(np.float64), 1.0, 1.0
        else:
            gq, hq, sg, sh = quantize(lam, hes, levels)
        # features of this tree: the n_used smallest hashes
        hv = np.array([splitmix64(splitmix64(p["seed"] + 1000003 * it) ^ f) for f in range(F)], dtype=np.uint64)
        used = np.zeros(F, dtype=bool)
        used[np.argsort(hv, kind="stable")[:n_used]] = True
        leaf_rows = {0: np.arange(n)}
        hists = {}

        def hist_of(rows):
            h = np.zeros((F, 256, 3), dtype=np.float64 if mode == "float" else np.int64)
            for f in range(F):
                np.add.at(h[f, :, 0], Xb[rows, f], gq[rows])      # (sequential: f64 sums in row order in float mode)
                np.add.at(h[f, :, 1], Xb[rows, f], hq[rows])
                np.add.at(h[f, :, 2], Xb[rows, f], 1)
            return h

        hists[0] = hist_of(leaf_rows[0])
        cand = {0: _best_split(hists[0], nb, used, sg, sh, p, nanbin)}
        nodes = []          # (feature, bin, left, right, gain, count, Gq, Hq) ; children >=0 node, <0 ~leaf
        leaf_parent = {0: None}
        n_leaves = 1
        while n_leaves < p["num_leaves"]:
            live = [(v[0], -l) for l, v in cand.items() if v is not None]
            if not live:
                break
            _, negl = max(live)                         # largest gain; ties -> smallest leaf index
            leaf = -negl
            gain, f, b, _, _, _, dl = cand.pop(leaf)
            rows = leaf_rows.pop(leaf)
            xb = Xb[rows, f]
            go_left = np.where(xb == nanbin[f], dl, xb <= b) if nanbin[f] >= 0 else xb <= b
            lrows, rrows = rows[go_left], rows[~go_left]
            node = len(nodes)
            tq = hists[leaf][f, :nb[f]].sum(0)
            nodes.append([f, b, ~leaf, ~n_leaves, gain, len(rows), tq[0], tq[1], bool(dl)])
            par = leaf_parent[leaf]
            if par is not None:
                nodes[par[0]][2 + par[1]] = node
            new_leaf = n_leaves
            leaf_rows[leaf], leaf_rows[new_leaf] = lrows, rrows
            leaf_parent[leaf], leaf_parent[new_leaf] = (node, 0), (node, 1)
            ph = hists.pop(leaf)
            small, big = (leaf, new_leaf) if len(lrows) <= len(rrows) else (new_leaf, leaf)
            hists[small] = hist_of(leaf_rows[small])
            hists[big] = ph - hists[small]
            n_leaves += 1
            for l in (leaf, new_leaf):
                cand[l] = _best_split(hists[l], nb, used, sg, sh, p, nanbin) if len(leaf_rows[l]) >= 2 * p["min_child_samples"] else None
        # leaf values from the (quantised) sums
        leaf_value = np.zeros(n_leaves)
        leaf_count = np.zeros(n_leaves, dtype=np.int64)
        for l, rows in leaf_rows.items():
            G = gq[rows].sum() / sg if sg else 0.0
            H = hq[rows].sum() / sh if sh else 0.0
            leaf_value[l] = -_thr_l1(G, p["reg_alpha"]) / (H + p["reg_lambda"]) * p["learning_rate"]
            leaf_count[l] = len(rows)
            scores[rows] += leaf_value[l]
        tree = dict(num_leaves=n_leaves, split_feature=np.array([nd[0] for nd in nodes], np.int64),
                    split_bin=np.array([nd[1] for nd in nodes], np.int64),
                    threshold=np.array([min(bounds[nd[0]][nd[1]], 1.7976931348623157e308) for nd in nodes], np.float64),
                    default_left=np.array([nd[8] for nd in nodes], bool),
                    nanbin=np.array([nanbin[nd[0]] for nd in nodes], np.int64),
                    left_child=np.array([nd[2] for nd in nodes], np.int64),
                    right_child=np.array([nd[3] for nd in nodes], np.int64),
                    split_gain=np.array([nd[4] for nd in nodes], np.float64), leaf_value=leaf_value,
                    leaf_count=leaf_count,
                    # bit 1 = default_left, bits 2-3 = missing type (2 = NaN); no missing bin: LightGBM's usual 2
                    decision_type=np.array([(8 | (2 if nd[8] else 0)) if nanbin[nd[0]] >= 0 else 2 for nd in nodes], np.int64),
                    num_cat=0, shrinkage=p["learning_rate"])
        trees.append(tree)
        rec = {"train": ndcg_at(scores, y, groups, p["eval_at"], gain_tab)}
        if Xvb is not None:
            vscores += predict_tree_binned(tree, Xvb)
            rec["valid"] = ndcg_at(vscores, yv, groups_v, p["eval_at"], gain_tab)
            cur = rec["valid"]
            # lightgbm.early_stopping (callback.py, first_metric_only=False): per-metric best score / best iteration;
            # the metrics are visited in order, the first whose patience ran out stops the run and names
            # best_iteration; a run that reaches n_estimators reports metric 0's best iteration
            stop = False
            if best_val is None:
                best_val, best_itr, best_it = list(cur), [it + 1] * len(cur), it + 1
            else:
                for t, v in enumerate(cur):
                    if v > best_val[t]:
                        best_val[t], best_itr[t] = v, it + 1
                    elif it + 1 - best_itr[t] >= p["early_stopping_rounds"]:
                        stop, best_it = True, best_itr[t]
                        break
                if not stop:
                    best_it = best_itr[0]
            history.append(rec)
            if stop:
                break
        else:
            history.append(rec)
            best_it = it + 1
    if Xvb is not None and 0 < best_it < len(trees):
        trees = trees[:best_it]     # Booster.predict / save_model default to best_iteration after early stopping
    names = feature_names or [f"Column_{i}" for i in range(F)]
    return dict(feature_names=names, max_feature_idx=F - 1, num_class=1, num_tree_per_iteration=1, average_output=False,
                objective="lambdarank", trees=trees, best_iteration=best_it, history=history, bounds=bounds)


def predict_tree_binned(tree: Dict, Xb: np.ndarray) -> np.ndarray:
    out = np.zeros(Xb.shape[0])
    if tree["num_leaves"] == 1:
        return out + tree["leaf_value"][0]
    node = np.zeros(Xb.shape[0], dtype=np.int64)
    active = np.ones(Xb.shape[0], dtype=bool)
    while active.any():
        idx = np.nonzero(active)[0]
        nd = node[idx]
        xb = Xb[idx, tree["split_feature"][nd]]
        left = xb <= tree["split_bin"][nd]
        if "nanbin" in tree:
            left = np.where(xb == tree["nanbin"][nd], tree["default_left"][nd], left)
        nxt = np.where(left, tree["left_child"][nd], tree["right_child"][nd])
        node[idx] = nxt
        done = nxt < 0
        out[idx[done]] = tree["leaf_value"][~nxt[done]]
        active[idx[done]] = False
    return out
