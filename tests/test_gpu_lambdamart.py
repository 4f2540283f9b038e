"""GPU LambdaMART trainer (csrc/gbdt_train.hip; replaces the body of src/models/ranker.py:52-155) against its NumPy
restatement oracle/lambdamart_np.py: lightgbm is not importable here, so this is *parity unpinned vs lightgbm, pinned
to the oracle*.  Gradients enter the histograms as integers, so tree structure, split features and thresholds must be
IDENTICAL and leaf values / metrics equal to 1e-9; the learned model must also rank: NDCG rises well above the
untrained level on a set with a learnable signal, and early stopping on a validation set reports a best iteration."""
import numpy as np
import pandas as pd
import pytest

from oracle import gbdt_np as G
from oracle import lambdamart_np as LM

pytestmark = pytest.mark.gpu


def _ranking_set(rng, n_q, docs, F, noise=0.7, grades=2):
    rows, labels, qid = [], [], []
    w = np.random.RandomState(1234 + F).randn(F)      # the same relevance model for every set of this width
    for q in range(n_q):
        n = rng.randint(docs[0], docs[1] + 1)
        X = rng.randn(n, F).astype(np.float32)
        X[:, 3] = np.round(X[:, 3], 1)                       # a feature with few distinct values
        X[:, 5] = (rng.rand(n) < 0.3).astype(np.float32)     # a binary feature
        rel = X @ w + noise * rng.randn(n)
        cut = np.quantile(rel, [0.7, 0.9][:grades - 1]) if grades > 1 else []
        lab = np.zeros(n, np.float32)
        for c in cut:
            lab += (rel > c)
        rows.append(X); labels.append(lab); qid += [q] * n
    X = np.concatenate(rows)
    df = pd.DataFrame(X, columns=[f"f{i}" for i in range(F)])
    df["label"] = np.concatenate(labels)
    df["query_id"] = qid
    return df


def _model_from_text(text):
    return G.parse_text_model(text)


def test_trees_identical_to_oracle_and_metrics_match():
    from recommendit_amd import LightGBMRanker
    rng = np.random.RandomState(0)
    F = 12
    tr = _ranking_set(rng, 150, (20, 60), F, grades=3)
    va = _ranking_set(rng, 40, (20, 60), F, grades=3)
    cols = [f"f{i}" for i in range(F)]
    rk = LightGBMRanker(num_leaves=15, n_estimators=6, learning_rate=0.1, eval_at=[5, 10])
    res = rk.train(tr, cols, valid_df=va, backend="hip")
    m = _model_from_text(rk._text)
    groups = tr.groupby("query_id", sort=False).size().values
    gv = va.groupby("query_id", sort=False).size().values
    o = LM.train(tr[cols].values.astype(np.float32), tr["label"].values, groups,
                 dict(num_leaves=15, n_estimators=6, learning_rate=0.1, eval_at=[5, 10]),
                 Xv=va[cols].values.astype(np.float32), yv=va["label"].values, groups_v=gv, feature_names=cols)
    # with a validation set the served model is cut at best_iteration (lightgbm: Booster.predict / save_model default)
    assert len(m["trees"]) == len(o["trees"]) == o["best_iteration"] == rk.best_iteration <= 6
    assert len(o["history"]) == len(res["valid"]["ndcg@5"]) == 6
    for t, (a, b) in enumerate(zip(m["trees"], o["trees"])):
        assert a["num_leaves"] == b["num_leaves"], t
        np.testing.assert_array_equal(a["split_feature"], b["split_feature"], err_msg=f"tree {t}")
        np.testing.assert_array_equal(a["left_child"], b["left_child"], err_msg=f"tree {t}")
        np.testing.assert_array_equal(a["right_child"], b["right_child"], err_msg=f"tree {t}")
        np.testing.assert_array_equal(a["threshold"], b["threshold"], err_msg=f"tree {t}")
        np.testing.assert_allclose(a["leaf_value"], b["leaf_value"], rtol=1e-9, atol=1e-12, err_msg=f"tree {t}")
    for name, idx in (("train", "train"), ("valid", "valid")):
        for t, k in enumerate((5, 10)):
            ref = [h[idx][t] for h in o["history"]]
            np.testing.assert_allclose(res[name][f"ndcg@{k}"], ref, rtol=0, atol=1e-9)
    # the text model is what the predictor serves: scores = sum of the oracle's trees
    Xq = va[cols].values.astype(np.float32)
    pred = rk.predict(va)
    np.testing.assert_allclose(pred, G.predict_raw(m, Xq), rtol=0, atol=1e-12)


def test_training_learns_and_early_stopping_reports_best_iteration():
    from recommendit_amd import LightGBMRanker
    rng = np.random.RandomState(1)
    F = 20
    tr = _ranking_set(rng, 400, (30, 120), F, noise=0.5)
    va = _ranking_set(rng, 100, (30, 120), F, noise=0.5)
    cols = [f"f{i}" for i in range(F)]
    rk = LightGBMRanker(num_leaves=31, n_estimators=60, learning_rate=0.1)
    res = rk.train(tr, cols, valid_df=va, backend="hip")
    v10 = res["valid"]["ndcg@10"]
    assert len(v10) >= 10 and v10[-1] > v10[0] + 0.02 and max(v10) > 0.6, (v10[0], v10[-1])
    assert 1 <= rk.best_iteration <= len(v10)
    assert rk.model.num_trees() == rk.best_iteration      # trees past the best iteration are neither served nor saved
    imp = rk.feature_importance()
    assert len(imp) == F and sum(imp.values()) > 0
    # save / load round trip through the LightGBM text format
    import tempfile, os
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "r.lgbm")
        rk.save(p)
        rk2 = LightGBMRanker.load(p)
        np.testing.assert_array_equal(rk2.predict(va), rk.predict(va))


def test_early_stop_serves_best_iteration_of_the_metric_that_stopped():
    """after lgb.early_stopping the reference scores and saves best_iteration trees, and the best
    iteration is that of the metric whose patience ran out (not always metric 0).  Pinned to oracle/lambdamart_np
    (lightgbm itself is not importable here: parity unpinned)."""
    from recommendit_amd import LightGBMRanker
    rng = np.random.RandomState(3)
    F = 6
    tr = _ranking_set(rng, 30, (10, 25), F, noise=2.0, grades=3)
    va = _ranking_set(rng, 20, (10, 25), F, noise=2.0, grades=3)
    cols = [f"f{i}" for i in range(F)]
    prm = dict(num_leaves=7, n_estimators=80, learning_rate=0.3, eval_at=[5, 10])
    rk = LightGBMRanker(**prm)
    res = rk.train(tr, cols, valid_df=va, backend="hip")
    o = LM.train(tr[cols].values.astype(np.float32), tr["label"].values, tr.groupby("query_id", sort=False).size().values,
                 prm, Xv=va[cols].values.astype(np.float32), yv=va["label"].values,
                 groups_v=va.groupby("query_id", sort=False).size().values, feature_names=cols)
    rounds = len(res["valid"]["ndcg@5"])
    assert rounds == len(o["history"]) < 80                       # the stop fired
    assert rk.best_iteration == o["best_iteration"] < rounds
    v5, v10 = res["valid"]["ndcg@5"], res["valid"]["ndcg@10"]
    stopped_by = [m for m in (v5, v10) if rounds - (int(np.argmax(m)) + 1) >= 30]
    assert stopped_by and rk.best_iteration == int(np.argmax(stopped_by[0])) + 1
    assert rk.model.num_trees() == rk.best_iteration == len(o["trees"])
    m = _model_from_text(rk._text)
    np.testing.assert_allclose(rk.predict(va), G.predict_raw(m, va[cols].values.astype(np.float32)), rtol=0, atol=1e-12)
    np.testing.assert_allclose(rk.predict(va), G.predict_raw(o, va[cols].values.astype(np.float32)), rtol=1e-9, atol=1e-9)


def test_large_query_groups_and_many_bins():
    """a query with thousands of documents (ML-1M users with ~2000 positives x 5) and continuous features with more than
    255 distinct values (equal-frequency cuts)"""
    from recommendit_amd import LightGBMRanker
    rng = np.random.RandomState(2)
    F = 8
    tr = _ranking_set(rng, 6, (3000, 9000), F, noise=0.5)
    cols = [f"f{i}" for i in range(F)]
    rk = LightGBMRanker(num_leaves=15, n_estimators=3, learning_rate=0.1, eval_at=[10])
    res = rk.train(tr, cols, backend="hip")
    groups = tr.groupby("query_id", sort=False).size().values
    o = LM.train(tr[cols].values.astype(np.float32), tr["label"].values, groups,
                 dict(num_leaves=15, n_estimators=3, learning_rate=0.1, eval_at=[10]), feature_names=cols)
    m = _model_from_text(rk._text)
    for a, b in zip(m["trees"], o["trees"]):
        np.testing.assert_array_equal(a["split_feature"], b["split_feature"])
        np.testing.assert_array_equal(a["threshold"], b["threshold"])
        np.testing.assert_allclose(a["leaf_value"], b["leaf_value"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res["train"]["ndcg@10"], [h["train"][0] for h in o["history"]], atol=1e-9)


def _xy(df, cols):
    return (df[cols].values.astype(np.float32), df["label"].values, df.groupby("query_id", sort=False).size().values)


def _same_trees(m, o, leaf_rtol=1e-9, exact_thr=True):
    assert len(m["trees"]) == len(o["trees"])
    for t, (a, b) in enumerate(zip(m["trees"], o["trees"])):
        assert a["num_leaves"] == b["num_leaves"], t
        np.testing.assert_array_equal(a["split_feature"], b["split_feature"], err_msg=f"tree {t}")
        np.testing.assert_array_equal(a["left_child"], b["left_child"], err_msg=f"tree {t}")
        np.testing.assert_array_equal(a["right_child"], b["right_child"], err_msg=f"tree {t}")
        np.testing.assert_array_equal(a["decision_type"], b["decision_type"], err_msg=f"tree {t}")
        if exact_thr:
            np.testing.assert_array_equal(a["threshold"], b["threshold"], err_msg=f"tree {t}")
        np.testing.assert_allclose(a["leaf_value"], b["leaf_value"], rtol=leaf_rtol, atol=1e-12, err_msg=f"tree {t}")


def test_float_histogram_mode_vs_its_oracle_branches():
    """hist_dtype="float" (2^-40 fixed point, order-independent) is (a) bit-identical to the oracle's int40 restatement
    and (b) equal -- same splits, leaf values to 1e-7 -- to the oracle's FLOAT branch: unquantised gradients, f64
    histogram sums in row order, thresholds found by LightGBM's two sequential scans (written from the published
    FeatureHistogram::FindBestThreshold, not from the kernel).  lightgbm itself is absent: parity unpinned."""
    from recommendit_amd import LightGBMRanker
    rng = np.random.RandomState(5)
    F = 10
    tr = _ranking_set(rng, 120, (20, 60), F, grades=3)
    cols = [f"f{i}" for i in range(F)]
    prm = dict(num_leaves=15, n_estimators=5, learning_rate=0.1, eval_at=[5, 10])
    rk = LightGBMRanker(**prm)
    res = rk.train(tr, cols, backend="hip", hist_dtype="float")
    m = _model_from_text(rk._text)
    X, y, g = _xy(tr, cols)
    _same_trees(m, LM.train(X, y, g, dict(prm, hist_dtype="int40"), feature_names=cols))
    of = LM.train(X, y, g, dict(prm, hist_dtype="float"), feature_names=cols)
    _same_trees(m, of, leaf_rtol=1e-7)
    np.testing.assert_allclose(res["train"]["ndcg@10"], [h["train"][1] for h in of["history"]], atol=1e-9)
    # and it is not the 2^20 mode: the leaf values differ from the quantised run at the 1e-6 level
    rk20 = LightGBMRanker(**prm)
    rk20.train(tr, cols, backend="hip")
    m20 = _model_from_text(rk20._text)
    assert any(not np.array_equal(a["leaf_value"], b["leaf_value"]) for a, b in zip(m["trees"], m20["trees"]))


@pytest.mark.parametrize("order", ["low", "lightgbm"])
@pytest.mark.parametrize("hist", ["int20", "float"])
def test_missing_values_learn_a_default_direction(order, hist):
    """use_missing=True with 15 % NaN: missing bin + default direction per node, trees identical to the oracle's
    sequential-scan branch in both tie orders; the text model carries missing type NaN (decision_type 8 / 10) and the
    HIP predictor routes NaN rows by it (scores equal to the oracle's walk of the same model)"""
    from recommendit_amd import LightGBMRanker
    rng = np.random.RandomState(6)
    F = 9
    tr = _ranking_set(rng, 100, (20, 50), F, grades=3)
    cols = [f"f{i}" for i in range(F)]
    X = tr[cols].values.astype(np.float32)
    miss = rng.rand(*X.shape) < 0.15
    miss[:, 2] = False                                   # one feature without any NaN: keeps decision_type 2
    X[miss] = np.nan
    X[:, 4] = np.where(rng.rand(len(X)) < 0.5, np.nan, X[:, 4])       # informative missingness
    tr[cols] = X
    prm = dict(num_leaves=15, n_estimators=4, learning_rate=0.1, eval_at=[5])
    rk = LightGBMRanker(**prm)
    rk.train(tr, cols, backend="hip", use_missing=True, split_order=order, hist_dtype=hist)
    m = _model_from_text(rk._text)
    Xf, y, g = _xy(tr, cols)
    o = LM.train(Xf, y, g, dict(prm, use_missing=True, split_order=order, hist_dtype="int40" if hist == "float" else hist),
                 feature_names=cols)
    _same_trees(m, o)
    dts = np.concatenate([t["decision_type"] for t in m["trees"]])
    assert set(dts.tolist()) <= {2, 8, 10} and (dts == 8).any() and (dts == 10).any()
    np.testing.assert_allclose(rk.predict(tr), G.predict_raw(m, Xf), rtol=0, atol=1e-12)
    # without use_missing the same data trains as if NaN were 0.0 and every node is the plain decision_type 2
    rk0 = LightGBMRanker(**prm)
    rk0.train(tr, cols, backend="hip")
    assert set(np.concatenate([t["decision_type"] for t in _model_from_text(rk0._text)["trees"]]).tolist()) == {2}


def test_lightgbm_tie_order_keeps_the_highest_threshold_of_an_empty_run():
    """few rows per leaf => runs of empty bins => equal gains: split_order="lightgbm" must pick the highest threshold of
    the run (FindBestThreshold scans right-to-left with a strict '>'), "low" the lowest; both equal their oracle branch"""
    from recommendit_amd import LightGBMRanker
    rng = np.random.RandomState(7)
    F = 6
    tr = _ranking_set(rng, 40, (15, 30), F, grades=3)
    cols = [f"f{i}" for i in range(F)]
    prm = dict(num_leaves=15, n_estimators=3, learning_rate=0.1, eval_at=[5])
    X, y, g = _xy(tr, cols)
    thr = {}
    for order in ("low", "lightgbm"):
        rk = LightGBMRanker(**prm)
        rk.train(tr, cols, backend="hip", split_order=order)
        m = _model_from_text(rk._text)
        _same_trees(m, LM.train(X, y, g, dict(prm, split_order=order), feature_names=cols))
        thr[order] = np.concatenate([t["threshold"] for t in m["trees"]])
    assert (thr["lightgbm"][: len(thr["low"])] >= thr["low"][: len(thr["lightgbm"])]).mean() > 0.5
    assert not np.array_equal(thr["lightgbm"], thr["low"])
