"""GPU parity: HIP forest traversal vs the GBDT oracle (float64; tolerance 1e-12 abs -- only the
summation order over tree chunks differs)."""
import numpy as np
import pandas as pd
import pytest
import torch

from oracle import gbdt_np as G

pytestmark = pytest.mark.gpu


def test_tiny_forest_hand_computed_on_gpu(golden_dir):
    from recommendit_amd import LightGBMRanker
    from tests.test_ranker_oracle import ROWS, EXPECTED
    r = LightGBMRanker.load(str(golden_dir / "tiny_forest.txt"))
    assert r.feature_names == ["fa", "fb", "fc"] and r.n_features == 3
    df = pd.DataFrame(ROWS, columns=["fa", "fb", "fc"])
    df["extra"] = 1.0
    out = r.predict(df[["extra", "fc", "fb", "fa"]])  # column order must not matter (ranker.py:173)
    assert out.dtype == np.float64
    np.testing.assert_allclose(out, EXPECTED, rtol=0, atol=1e-12)
    imp = r.feature_importance("split")
    assert imp == {"fb": 2, "fa": 1, "fc": 1}
    assert r.feature_importance("gain")["fa"] == 10.0
    assert r.model_info()["n_features"] == 3


@pytest.mark.parametrize("n_trees,n_leaves,n_feat,n", [(500, 63, 50, 500), (37, 15, 10, 1), (3, 2, 128, 70),
                                                        (200, 200, 50, 1000)])
def test_random_forest_vs_oracle(tmp_path, n_trees, n_leaves, n_feat, n):
    from recommendit_amd import LightGBMRanker
    model = G.random_forest_model(n_trees, n_leaves, n_feat, seed=n_trees)
    rng = np.random.RandomState(1)
    # sprinkle the non-default decision types over the forest
    for t in model["trees"][::3]:
        t["decision_type"][::2] = 10  # NaN-missing, default left
    for t in model["trees"][1::3]:
        t["decision_type"][::2] = 4   # zero-missing, default right
    p = tmp_path / "forest.lgbm"
    p.write_text(G.write_text_model(model))
    r = LightGBMRanker.load(str(p))
    X = rng.randn(n, n_feat).astype(np.float32)
    X[rng.rand(n, n_feat) < 0.05] = np.nan
    X[rng.rand(n, n_feat) < 0.05] = 0.0
    got = r.model.predict(X)
    exp = G.predict_raw(model, X)
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-12)
    # bitwise reproducible
    np.testing.assert_array_equal(got, r.model.predict(X))


def test_categorical_split(tmp_path):
    from recommendit_amd import LightGBMRanker
    tree = dict(num_leaves=2, num_cat=1, split_feature=np.array([0]), threshold=np.array([0.0]),
                decision_type=np.array([1]), left_child=np.array([-1]), right_child=np.array([-2]),
                leaf_value=np.array([1.5, -2.5]), cat_boundaries=np.array([0, 2]),
                cat_threshold=np.array([(1 << 3) | (1 << 7), 1 << 1]), shrinkage=1.0)
    model = dict(feature_names=["c"], trees=[tree])
    p = tmp_path / "cat.lgbm"
    p.write_text(G.write_text_model(model))
    r = LightGBMRanker.load(str(p))
    X = np.array([[3.0], [7.0], [33.0], [4.0], [-1.0], [np.nan], [64.0]], dtype=np.float32)
    exp = np.array([1.5, 1.5, 1.5, -2.5, -2.5, -2.5, -2.5])
    np.testing.assert_array_equal(G.predict_raw(G.parse_text_model(p.read_text()), X), exp)
    np.testing.assert_array_equal(r.model.predict(X), exp)


def test_errors(tmp_path):
    from recommendit_amd import LightGBMRanker
    with pytest.raises(RuntimeError, match="not trained"):
        LightGBMRanker().predict(pd.DataFrame({"a": [1.0]}))
    with pytest.raises(FileNotFoundError):
        LightGBMRanker.load(str(tmp_path / "nope.lgbm"))
    (tmp_path / "bad.lgbm").write_text("this is not a model\n")
    with pytest.raises(RuntimeError):
        LightGBMRanker.load(str(tmp_path / "bad.lgbm"))


@pytest.mark.parametrize("n_trees,n_leaves,n_feat,n", [(500, 63, 50, 128000), (70, 64, 64, 5000), (129, 31, 7, 2048),
                                                        (500, 63, 50, 2047), (65, 2, 3, 4100)])
def test_forest_walk_large_batches_and_edge_values_vs_oracle(tmp_path, n_trees, n_leaves, n_feat, n):
    """plain numerical forests at serving batch sizes (128 000 candidates = 256 requests x 500) with the values a walk can
    get wrong: NaN -> 0.0, x == threshold goes left, +-inf features.  (Round 3 also tried 4-byte records over pre-binned
    features -- 64-tree chunks, 4 walks in flight per lane; same 488 M candidates/s, dropped: DESIGN.md §9.)"""
    from recommendit_amd import LightGBMRanker
    model = G.random_forest_model(n_trees, n_leaves, n_feat, seed=n_trees + n_leaves)
    p = tmp_path / "forest.lgbm"
    p.write_text(G.write_text_model(model))
    r = LightGBMRanker.load(str(p))
    rng = np.random.RandomState(2)
    nn = min(n, 20000)                                   # the oracle walks in NumPy: compare on a sample of the rows
    X = rng.randn(n, n_feat).astype(np.float32)
    X[rng.rand(n, n_feat) < 0.03] = np.nan
    X[rng.rand(n, n_feat) < 0.01] = np.inf
    X[rng.rand(n, n_feat) < 0.01] = -np.inf
    thr = np.concatenate([t["threshold"] for t in model["trees"] if len(t["threshold"])] or [np.zeros(1)])
    hit = rng.rand(n, n_feat) < 0.05                     # exact threshold values (as float32): the <= must hold
    X[hit] = rng.choice(thr, size=int(hit.sum())).astype(np.float32)
    got = r.model.predict(X)
    sel = rng.choice(n, size=nn, replace=False) if nn < n else np.arange(n)
    np.testing.assert_allclose(got[sel], G.predict_raw(model, X[sel]), rtol=0, atol=1e-12)
    np.testing.assert_array_equal(got, r.model.predict(X))          # bitwise reproducible
    np.testing.assert_allclose(r.model.predict(X[:1500]), got[:1500], rtol=0, atol=1e-12)   # batch-size independent
