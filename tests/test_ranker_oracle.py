"""CPU: the GBDT oracle against a hand-written LightGBM-format model with hand-computed scores
(lightgbm itself is absent: parity unpinned beyond this fixture)."""
import numpy as np

from oracle import gbdt_np as G

# rows and expected raw scores, computed by hand from tests/golden/tiny_forest.txt
ROWS = np.array([[0.3, -2.0, 0.0], [0.3, 0.0, 5.0], [0.7, np.nan, np.nan], [0.5, -1.0, -3.0]], dtype=np.float32)
EXPECTED = np.array([0.1 - 1 + 0.05 + 20, 0.3 + 1 + 0.05 + 20, 0.2 - 1 + 0.05 + 10, 0.1 - 1 + 0.05 + 20])


def test_tiny_forest_hand_computed(golden_dir):
    model = G.parse_text_model((golden_dir / "tiny_forest.txt").read_text())
    assert model["feature_names"] == ["fa", "fb", "fc"] and len(model["trees"]) == 4
    np.testing.assert_allclose(G.predict_raw(model, ROWS), EXPECTED, rtol=0, atol=1e-12)


def test_text_round_trip_of_synthetic_forest():
    m = G.random_forest_model(n_trees=7, n_leaves=9, n_features=5, seed=3)
    m2 = G.parse_text_model(G.write_text_model(m))
    X = np.random.RandomState(0).randn(64, 5).astype(np.float32)
    np.testing.assert_array_equal(G.predict_raw(m, X), G.predict_raw(m2, X))
