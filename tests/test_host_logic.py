"""CPU: host-side logic of the product package (no GPU, no oracle needed except as cross-check)."""
import math

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import metrics_np as OM
from oracle import ranking_features_np as RF
from oracle import two_tower_np as O


def test_cosine_lr_matches_torch_scheduler():
    from recommendit_amd.trainer import cosine_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10)
    for e in range(10):
        assert abs(cosine_lr(1e-3, e, 10) - opt.param_groups[0]["lr"]) < 1e-12
        assert abs(cosine_lr(1e-3, e, 10) - O.cosine_lr(1e-3, e, 10)) < 1e-15
        opt.step(); sch.step()


def test_metrics_match_oracle_and_reference_known_answers():
    from recommendit_amd import metrics as M
    rng = np.random.RandomState(0)
    for _ in range(50):
        rec = rng.permutation(30)[:20].tolist()
        rel = rng.permutation(30)[: rng.randint(0, 6)].tolist()
        for k in (5, 10, 20):
            assert M.ndcg_at_k(rec, rel, k) == OM.ndcg_at_k(rec, rel, k)
            assert M.recall_at_k(rec, rel, k) == OM.recall_at_k(rec, rel, k)
        assert M.mrr(rec, rel) == OM.mrr(rec, rel)
    assert abs(M.ndcg_at_k([1, 2, 3, 4, 5], [1, 2, 3], 3) - 1.0) < 1e-12      # reference tests/test_models.py:372-378
    res = M.evaluate_model({1: [1, 2, 3], 2: [9, 8, 7], 3: [5]}, {1: [1], 2: [7], 3: []}, [1, 3])
    assert res["n_users"] == 3 and abs(res["ndcg@1"] - 0.5) < 1e-12 and abs(res["mrr"] - (1.0 + 1 / 3) / 2) < 1e-12


def test_synthetic_ml1m_shape_and_determinism():
    from recommendit_amd.synthetic import ml1m_like
    r1, m1, g1 = ml1m_like(n_users=300, n_item_ids=260, n_catalog=250, n_ratings=20000, seed=3)
    r2, m2, g2 = ml1m_like(n_users=300, n_item_ids=260, n_catalog=250, n_ratings=20000, seed=3)
    assert r1.equals(r2) and m1.equals(m2) and np.array_equal(g1, g2)
    assert r1["user_id"].nunique() == 300 and set(r1["rating"].unique()) <= {1, 2, 3, 4, 5}
    assert r1.groupby("user_id").size().min() >= 20 and len(m1) == 250 and g1.shape == (261, 18)
    assert not r1.duplicated(["user_id", "item_id"]).any()
    assert (r1.groupby("user_id")["timestamp"].apply(lambda s: s.is_monotonic_increasing)).all()


def test_dataset_host_path_follows_reference_semantics():
    from recommendit_amd.synthetic import ml1m_like
    from recommendit_amd.train_embeddings import UserItemDataset, build_item_genre_dict
    ratings, movies, gm = ml1m_like(n_users=100, n_item_ids=120, n_catalog=110, n_ratings=5000, seed=1)
    gd = build_item_genre_dict(movies)
    assert all(np.array_equal(gd[i], gm[i]) for i in movies["item_id"])
    ds = UserItemDataset(ratings, gd, sorted(movies["item_id"].tolist()))
    assert len(ds) == int((ratings["rating"] >= 4).sum())          # train_embeddings.py:43-45
    rated = ratings.groupby("user_id")["item_id"].apply(set).to_dict()
    np.random.seed(0)
    for i in range(0, len(ds), max(1, len(ds) // 40)):
        u, p, gp, n, gn = ds[i]
        assert int(n) not in rated[int(u)] and int(n) in gd and int(p) in rated[int(u)]
        assert gp.shape == (18,) and gp.dtype == torch.float32 and u.dtype == torch.long


def test_gpu_feature_store_tables_match_reference_defaults():
    from recommendit_amd.recommender import GpuFeatureStore, feature_columns
    st = GpuFeatureStore(3, 4)
    assert st.user.shape == (4, 24) and st.item.shape == (5, 23)
    assert st.user[2, :6].tolist() == [3.5, 0.0, 0.5, 0.0, 0.3, 0.3]     # recommender.py:227-232
    assert st.item[1, :5].tolist() == [3.5, 0.0, 0.0, 0.0, 0.5]          # recommender.py:234-238
    st.set_user_features(2, {"avg_rating": 4.25, "genre_pref": [0.5] * 3})
    assert st.user[2, 0] == 4.25 and st.user[2, 6:9].tolist() == [0.5] * 3 and st.user[2, 9] == 0.0
    st.set_item_features(1, None)
    assert st.item[1, :5].tolist() == [3.5, 0.0, 0.0, 0.0, 0.5]
    assert feature_columns() == RF.feature_columns() and len(feature_columns()) == 50


def test_checkpoint_keys_and_hidden_dim_inference(tmp_path):
    from recommendit_amd import TwoTowerModel
    m = TwoTowerModel(12, 15, embed_dim=32, hidden_dim=64)
    m.save(str(tmp_path / "m.pt"))
    ck = torch.load(tmp_path / "m.pt", weights_only=True)
    assert sorted(ck) == ["embed_dim", "idx_to_item_id", "item_id_to_idx", "n_items", "n_users", "state_dict"]
    assert "user_tower.mlp.3.weight" in ck["state_dict"] and ck["state_dict"]["item_tower.mlp.0.weight"].shape == (64, 50)
    m2 = TwoTowerModel.load(str(tmp_path / "m.pt"))
    assert m2.hidden_dim == 64 and (m2.n_users, m2.n_items, m2.embed_dim) == (12, 15, 32)
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), m2.state_dict()[k].cpu())


def test_error_conventions_without_backend_calls(tmp_path):
    from recommendit_amd import FAISSIndex, LightGBMRanker
    with pytest.raises(RuntimeError, match="Index not built"):
        FAISSIndex(embed_dim=8).search(np.zeros(8, np.float32))
    with pytest.raises(RuntimeError, match="Index not built"):
        FAISSIndex(embed_dim=8).batch_search(np.zeros((1, 8), np.float32))
    assert FAISSIndex(embed_dim=8).stats() == {"status": "not built"}
    with pytest.raises(FileNotFoundError):
        FAISSIndex.load(str(tmp_path / "none.index"))
    with pytest.raises(RuntimeError, match="not trained"):
        LightGBMRanker().predict(pd.DataFrame({"a": [1.0]}))
    with pytest.raises(FileNotFoundError):
        LightGBMRanker.load(str(tmp_path / "none.lgbm"))
    assert LightGBMRanker().model_info() == {"status": "not trained"}
    assert LightGBMRanker().best_iteration == 0 and LightGBMRanker().n_features == 0


def test_feature_store_parquet_loader_matches_row_by_row_path(tmp_path):
    """GpuFeatureStore.from_parquet / load_all_features (vectorised) against the per-row dict route that restates the
    reference's RedisFeatureStore.load_all_features (src/features/feature_store.py:156-228), on files laid out as
    FeatureEngineer.save_features writes them (feature_engineering.py:376-406): scalars + genre_pref_<i> / genre_vec_<i>."""
    import pandas as pd
    from recommendit_amd.recommender import GpuFeatureStore, ITEM_SCALARS, USER_SCALARS
    rng = np.random.RandomState(0)
    nu, ni = 40, 55
    uids = rng.permutation(np.arange(1, nu + 1))[:33]                       # some users have no features
    iids = rng.permutation(np.arange(1, ni + 1))[:50]
    udf = pd.DataFrame({"user_id": uids, **{n: rng.rand(len(uids)) * 5 for n, _ in USER_SCALARS if n != "recency_score"},
                        **{f"genre_pref_{i}": rng.rand(len(uids)).astype(np.float32) for i in range(18)},
                        "rating_count": rng.randint(1, 99, len(uids))})    # extra column the ranker never reads
    idf = pd.DataFrame({"item_id": iids, "title": [f"Movie {i}" for i in iids],
                        **{n: rng.rand(len(iids)) * 3 for n, _ in ITEM_SCALARS},
                        **{f"genre_vec_{i}": (rng.rand(len(iids)) < 0.2).astype(np.float32) for i in range(18)}})
    udf.to_parquet(tmp_path / "user_features.parquet", index=False)
    idf.to_parquet(tmp_path / "item_features.parquet", index=False)
    st = GpuFeatureStore.from_parquet(str(tmp_path), n_users=nu, n_items=ni)
    ref = GpuFeatureStore(nu, ni)
    gp = [f"genre_pref_{i}" for i in range(18)]
    gv = [f"genre_vec_{i}" for i in range(18)]
    for _, row in pd.read_parquet(tmp_path / "user_features.parquet").iterrows():     # feature_store.py:183-191
        feat = {c: row[c] for c in udf.columns if c != "user_id" and c not in gp}
        feat["genre_pref"] = [float(row[c]) for c in gp]
        ref.set_user_features(int(row["user_id"]), feat)
    for _, row in pd.read_parquet(tmp_path / "item_features.parquet").iterrows():     # feature_store.py:211-221
        feat = {c: row[c] for c in idf.columns if c not in ("item_id", "title") and c not in gv}
        feat["genre_vector"] = [float(row[c]) for c in gv]
        ref.set_item_features(int(row["item_id"]), feat)
    np.testing.assert_array_equal(st.user, ref.user)
    np.testing.assert_array_equal(st.item, ref.item)
    assert (st.user[:, 2] == 0.5).all()                                       # absent column keeps its default
    missing = sorted(set(range(1, nu + 1)) - set(uids.tolist()))
    assert (st.user[missing, :6] == [d for _, d in USER_SCALARS]).all()
    # ids beyond the declared size grow the table; un-expanded array columns are accepted too
    st2 = GpuFeatureStore(3, 3)
    st2.load_all_features(pd.DataFrame({"user_id": [9], "avg_rating": [4.5], "genre_pref": [list(range(18))]}), None)
    assert st2.user.shape[0] == 10 and st2.user[9, 0] == 4.5 and st2.user[9, 6 + 17] == 17.0 and st2.user[5, 0] == 3.5


def test_faiss_io_roundtrip_and_hand_assembled_bytes(tmp_path):
    """recommendit_amd/faiss_io.py: writer -> reader round trip (full and sparse list tables) and a byte string
    assembled by hand from the published layout of faiss 1.7.x index_write.cpp (parity unpinned: no faiss here)."""
    import struct
    from recommendit_amd import faiss_io as F
    rng = np.random.RandomState(0)
    d, n, nlist = 8, 50, 6
    X = rng.randn(n, d).astype(np.float32)
    C = rng.randn(nlist, d).astype(np.float32)
    a = rng.randint(0, nlist, n).astype(np.int32)
    for assign in (a, np.where(a < 2, a, 0).astype(np.int32)):            # second: 2 of 6 lists non-empty -> "sprs"
        p = tmp_path / "i.faiss"
        F.write_ivf_flat(str(p), X, C, assign, nprobe=3)
        assert F.sniff(str(p)) == "faiss"
        r = F.read_index(str(p))
        assert (r["kind"], r["d"], r["ntotal"], r["nlist"], r["nprobe"], r["metric"]) == ("ivf", d, n, nlist, 3, 0)
        np.testing.assert_array_equal(r["vectors"], X)
        np.testing.assert_array_equal(r["centroids"], C)
        np.testing.assert_array_equal(r["assign"], assign)
    F.write_flat(str(tmp_path / "f.faiss"), X)
    r = F.read_index(str(tmp_path / "f.faiss"))
    assert r["kind"] == "flat"
    np.testing.assert_array_equal(r["vectors"], X)
    # hand-assembled IndexIVFFlat: d=2, 3 vectors, 2 lists (list 1 holds ids 2,0; list 0 holds id 1), array direct map
    hdr = lambda d_, nt: struct.pack("<iqqq?i", d_, nt, 1 << 20, 1 << 20, True, 0)
    cent = np.array([[1, 0], [0, 1]], np.float32)
    vec = np.array([[0.1, 0.9], [0.8, 0.2], [0.3, 0.7]], np.float32)
    blob = (b"IwFl" + hdr(2, 3) + struct.pack("<QQ", 2, 1)
            + b"IxFI" + hdr(2, 2) + struct.pack("<Q", 4) + cent.tobytes()
            + struct.pack("<b", 1) + struct.pack("<Q", 3) + np.array([5, 6, 7], np.int64).tobytes()
            + b"ilar" + struct.pack("<QQ", 2, 8) + b"full" + struct.pack("<Q", 2) + np.array([1, 2], np.uint64).tobytes()
            + vec[[1]].tobytes() + np.array([1], np.int64).tobytes()
            + vec[[2, 0]].tobytes() + np.array([2, 0], np.int64).tobytes())
    (tmp_path / "h.faiss").write_bytes(blob)
    r = F.read_index(str(tmp_path / "h.faiss"))
    np.testing.assert_array_equal(r["vectors"], vec)
    np.testing.assert_array_equal(r["assign"], [1, 0, 1])
    np.testing.assert_array_equal(r["centroids"], cent)
    with pytest.raises(F.FaissFormatError):
        (tmp_path / "t.faiss").write_bytes(blob[:-5]); F.read_index(str(tmp_path / "t.faiss"))
    (tmp_path / "x.bin").write_bytes(b"RIHIPIDX" + b"\0" * 64)
    assert F.sniff(str(tmp_path / "x.bin")) == "rihip"
