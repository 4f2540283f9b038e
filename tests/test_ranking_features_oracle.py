"""CPU: the ranking-feature oracle against the outputs of the reference's own _build_ranking_features (G8)."""
import json

import numpy as np

from oracle import ranking_features_np as RF


def load_g8(golden_dir):
    g = np.load(golden_dir / "g8_ranking_features.npz")
    meta = json.loads((golden_dir / "g8_inputs.json").read_text())
    return g, meta


def test_g8_columns_and_values(golden_dir):
    g, meta = load_g8(golden_dir)
    for m in meta:
        s = m["seed"]
        items = {int(k): v for k, v in m["items"].items()}
        cols = RF.build_ranking_features(m["user"], items, m["cand"])
        ref_cols = [str(c) for c in g[f"s{s}_columns"]]
        assert list(cols.keys()) == ref_cols
        got = np.stack([cols[c] for c in ref_cols], axis=1)
        np.testing.assert_array_equal(got, g[f"s{s}_values"])   # Python-float arithmetic: bit-exact
    assert set(RF.feature_columns()) == set(ref_cols) - {"item_id"} and len(RF.feature_columns()) == 50
