"""GPU parity: inner-product top-K vs the exact NumPy oracle, plus the reference's own property
tests for the FAISSIndex wrapper (tests/test_models.py:151-246 of the reference).

Float scores: |score - oracle| <= 2e-6 (unit vectors, exact-f32 fmaf chain vs float64).  Row sets:
bit-exact whenever the oracle's k-th/(k+1)-th gap exceeds that tolerance; near-ties are checked
through the score of the returned row instead.  Integer-valued inputs: bit-exact incl. tie-break."""
import tempfile
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import retrieval_np as R

pytestmark = pytest.mark.gpu
TOL = 2e-6


def _check_topk(scores, rows, Q, X, k):
    S = Q.astype(np.float64) @ X.astype(np.float64).T
    nq = Q.shape[0]
    assert scores.shape == (nq, k) and rows.shape == (nq, k)
    ref_sorted = -np.sort(-S, axis=1)[:, :k]
    np.testing.assert_allclose(scores, ref_sorted, atol=TOL, rtol=0)          # same score multiset
    assert (np.diff(scores, axis=1) <= 0).all()                               # descending
    got_true = np.take_along_axis(S, rows, axis=1)
    np.testing.assert_allclose(scores, got_true, atol=TOL, rtol=0)            # rows carry those scores
    for q in range(nq):
        assert len(set(rows[q].tolist())) == k                                # no duplicates
    # exact set equality where the boundary is not a near-tie
    ref_sc, ref_rows = R.topk_ip_exact(Q, X, min(k + 1, X.shape[0]))
    for q in range(nq):
        if X.shape[0] > k and ref_sc[q, k - 1] - ref_sc[q, k] > 4 * TOL:
            assert set(rows[q].tolist()) == set(ref_rows[q, :k].tolist()), q


def _index(X, exact=True, **kw):
    from recommendit_amd import FAISSIndex
    idx = FAISSIndex(embed_dim=X.shape[1], exact=exact, **kw)
    idx.build_ivf_index(X, list(range(1000, 1000 + X.shape[0])))
    return idx


@pytest.mark.parametrize("N,d,nq,k", [(500, 32, 7, 20), (3883, 64, 130, 500), (70000, 128, 33, 500),
                                      (200000, 64, 300, 500), (300000, 128, 64, 100)])
def test_bruteforce_topk_vs_oracle(N, d, nq, k):
    rng = np.random.RandomState(N)
    X, Q = fx.unit_rows(rng, N, d), fx.unit_rows(rng, nq, d)
    idx = _index(X)
    sc, ids = idx.batch_search(Q, k=k)
    _check_topk(sc, ids - 1000, Q, X, k)


def test_integer_inputs_bit_exact_with_ties():
    """Small-integer vectors: every dot product is exact in f32, many exact ties -> lowest row wins."""
    rng = np.random.RandomState(3)
    N, d, nq, k = 150000, 32, 40, 500
    X = rng.randint(-2, 3, size=(N, d)).astype(np.float32)
    Q = rng.randint(-2, 3, size=(nq, d)).astype(np.float32)
    from recommendit_amd import FAISSIndex
    idx = FAISSIndex(embed_dim=d, exact=True)
    idx.build_from_device(torch.from_numpy(X).cuda(), np.arange(N))          # no normalisation
    sc, rows = idx._search_device(torch.from_numpy(Q).cuda(), k)
    ref_sc, ref_rows = R.topk_ip_exact_f32(Q, X, k)
    np.testing.assert_array_equal(sc.cpu().numpy(), ref_sc)
    np.testing.assert_array_equal(rows.cpu().numpy(), ref_rows)


def test_all_identical_rows_take_the_exact_fallback():
    N, d, k = 70000, 32, 50
    X = np.tile(fx.unit_rows(np.random.RandomState(0), 1, d), (N, 1))
    Q = fx.unit_rows(np.random.RandomState(1), 11, d)
    from recommendit_amd import FAISSIndex
    idx = FAISSIndex(embed_dim=d, exact=True)
    idx.build_from_device(torch.from_numpy(X).cuda(), np.arange(N))
    sc, rows = idx._search_device(torch.from_numpy(Q).cuda(), k)
    np.testing.assert_array_equal(rows.cpu().numpy(), np.tile(np.arange(k), (11, 1)))
    assert (sc.cpu().numpy() == sc.cpu().numpy()[:, :1]).all()


def test_full_size_corpus_properties():
    """cfg3 size (N=1M, d=128, k=500): subset of queries checked exactly, all checked for order/validity."""
    rng = np.random.RandomState(1)
    N, d, nq, k = 1_000_000, 128, 256, 500
    X = torch.randn(N, d, generator=torch.Generator().manual_seed(1))
    X = (X / X.norm(dim=1, keepdim=True)).contiguous()
    Q = fx.unit_rows(rng, nq, d)
    from recommendit_amd import FAISSIndex
    idx = FAISSIndex(embed_dim=d, exact=True)
    idx.build_from_device(X.cuda(), np.arange(N))
    sc, rows = idx._search_device(torch.from_numpy(Q).cuda(), k)
    sc, rows = sc.cpu().numpy(), rows.cpu().numpy()
    assert (np.diff(sc, axis=1) <= 0).all() and (rows >= 0).all() and (rows < N).all()
    Xn = X.numpy()
    _check_topk(sc[:8], rows[:8], Q[:8], Xn, k)
    # idempotence: searching again gives bit-identical results
    sc2, rows2 = idx._search_device(torch.from_numpy(Q).cuda(), k)
    np.testing.assert_array_equal(rows, rows2.cpu().numpy())
    np.testing.assert_array_equal(sc, sc2.cpu().numpy())


# ---- the reference's FAISSIndex property tests, against the IVF path ------------------------------
class TestFAISSIndexReferenceProperties:
    EMBED_DIM = 32
    N_ITEMS = 500

    @pytest.fixture
    def built_index(self):
        from recommendit_amd import FAISSIndex
        np.random.seed(123)
        embeddings = np.random.randn(self.N_ITEMS, self.EMBED_DIM).astype(np.float32)
        embeddings = embeddings / np.linalg.norm(embeddings, axis=1, keepdims=True)
        item_ids = list(range(1, self.N_ITEMS + 1))
        index = FAISSIndex(embed_dim=self.EMBED_DIM, n_lists=10, n_probe=5)
        index.build_ivf_index(embeddings, item_ids)
        return index, embeddings, item_ids

    def test_index_built(self, built_index):
        index, _, _ = built_index
        assert index.index is not None and index.index.ntotal == self.N_ITEMS and index.index.is_ivf

    def test_search_returns_k_results_and_types(self, built_index):
        index, _, _ = built_index
        query = np.random.randn(self.EMBED_DIM).astype(np.float32)
        distances, retrieved_ids = index.search(query, k=20)
        assert len(distances) == 20 and len(retrieved_ids) == 20
        assert distances.dtype in [np.float32, np.float64]
        assert all(isinstance(i, (int, np.integer)) for i in retrieved_ids)
        assert (np.diff(distances) <= 0.01).all()

    def test_nearest_neighbor_is_self(self, built_index):
        index, embeddings, item_ids = built_index
        _, retrieved_ids = index.search(embeddings[42].copy(), k=5)
        assert item_ids[42] == retrieved_ids[0]

    def test_search_k_capped_and_short_results(self, built_index):
        index, _, _ = built_index
        query = np.random.randn(self.EMBED_DIM).astype(np.float32)
        d, ids = index.search(query, k=10000)
        assert 0 < len(ids) <= self.N_ITEMS and len(d) == len(ids)   # nprobe*N/nlist < k => fewer than k
        sc, bid = index.batch_search(query[None], k=10000)
        assert bid.shape == (1, self.N_ITEMS) and (bid[0, len(ids):] == -1).all()
        assert list(bid[0, :len(ids)]) == list(ids)

    def test_save_and_load(self, built_index):
        from recommendit_amd import FAISSIndex
        index, embeddings, _ = built_index
        with tempfile.TemporaryDirectory() as tmpdir:
            path = str(Path(tmpdir) / "faiss.index")
            index.save(path)
            assert Path(path).exists() and Path(path).with_suffix(".meta.pkl").exists()
            loaded = FAISSIndex.load(path)
            assert loaded.index.ntotal == self.N_ITEMS and loaded.embed_dim == self.EMBED_DIM
            d1, ids1 = index.search(embeddings[0].copy(), k=10)
            d2, ids2 = loaded.search(embeddings[0].copy(), k=10)
            assert list(ids1) == list(ids2)
            np.testing.assert_array_equal(d1, d2)
        with pytest.raises(FileNotFoundError):
            FAISSIndex.load("/nonexistent/faiss.index")

    def test_stats_and_nprobe(self, built_index):
        index, _, _ = built_index
        s = index.stats()
        assert s["n_vectors"] == self.N_ITEMS and s["embed_dim"] == self.EMBED_DIM and s["metric"] == "inner_product"
        index.set_n_probe(3)
        assert index.index.nprobe == 3 and index.n_probe == 3

    def test_unnormalized_query_handled(self, built_index):
        index, _, _ = built_index
        q = np.random.randn(self.EMBED_DIM).astype(np.float32) * 100
        _, ids1 = index.search(q, k=5)
        _, ids2 = index.search(q / np.linalg.norm(q), k=5)
        assert list(ids1) == list(ids2)

    def test_errors(self):
        from recommendit_amd import FAISSIndex
        with pytest.raises(RuntimeError, match="Index not built"):
            FAISSIndex(embed_dim=32).search(np.zeros(32, np.float32))
        with pytest.raises(AssertionError):
            FAISSIndex(embed_dim=32).build_ivf_index(np.zeros((4, 32), np.float64), [1, 2, 3, 4])
        with pytest.raises(AssertionError):
            FAISSIndex(embed_dim=32).build_ivf_index(np.zeros((4, 16), np.float32), [1, 2, 3, 4])


def test_ivf_full_probe_equals_exact_and_recall():
    rng = np.random.RandomState(8)
    N, d, nq, k = 20000, 64, 50, 100
    X, Q = fx.unit_rows(rng, N, d), fx.unit_rows(rng, nq, d)
    from recommendit_amd import FAISSIndex
    ivf = FAISSIndex(embed_dim=d, n_lists=50, n_probe=50)
    ivf.build_ivf_index(X, list(range(N)))
    sc, rows = ivf.batch_search(Q, k=k)
    _check_topk(sc, rows, Q, X, k)                       # nprobe == nlist  =>  exact
    ivf.set_n_probe(10)
    sc10, rows10 = ivf.batch_search(Q, k=k)
    ref_sc, ref_rows = R.topk_ip_exact(Q, X, k)
    recall = np.mean([len(set(rows10[q]) & set(ref_rows[q])) / k for q in range(nq)])
    assert recall > 0.5, recall
    # everything returned is a true inner product of a real row
    valid = rows10 >= 0
    S = Q.astype(np.float64) @ X.astype(np.float64).T
    np.testing.assert_allclose(sc10[valid], np.take_along_axis(S, np.where(valid, rows10, 0), 1)[valid], atol=TOL)


def test_two_precision_search_equals_all_f32():
    """bf16-MFMA filter + exact f32 re-score must return the same rows as the all-f32 search (completeness is proven
    per query; unproven queries take the exact fallback)."""
    from recommendit_amd import FAISSIndex, _lib
    rng = np.random.RandomState(11)
    N, d, nq, k = 250000, 128, 200, 500
    X, Q = fx.unit_rows(rng, N, d), fx.unit_rows(rng, nq, d)
    Q[:5] = X[:5]                                  # exact self-matches (score 1.0 on the boundary of the range)
    idx = FAISSIndex(embed_dim=d, exact=True)
    idx.build_ivf_index(X, list(range(N)))
    s2, r2 = idx.batch_search(Q, k=k)              # default: two-precision
    _lib.check(_lib.lib().rihip_ip_index_set_two_precision(idx.index._h, 0))
    s1, r1 = idx.batch_search(Q, k=k)              # all-f32
    _check_topk(s2, r2, Q, X, k)
    np.testing.assert_allclose(s2, s1, atol=1e-6, rtol=0)
    same = (r1 == r2).mean()
    assert same > 0.999, same                      # identical up to f32 near-ties re-ordered by summation order
    for q in range(nq):
        assert set(r1[q]) == set(r2[q]) or np.abs(s1[q, -1] - s1[q, -2]) < 1e-6


def test_wide_filter_equals_the_narrow_filter_and_all_f32(monkeypatch):
    """`RIHIP_FILTER_WIDE=1` sends batches of more than 512 queries at d = 128 to the 1 024-query LDS-DMA filter
    (`scan_bf16_wide_kernel`, an opt-in experiment: DESIGN.md section 9); it must hand the re-score the same survivors as
    the 256-query filter: identical results, also on a corpus that is not a whole number of 64-row stages, with a partly
    filled last query block and with self-matches."""
    from recommendit_amd import FAISSIndex, _lib
    rng = np.random.RandomState(21)
    N, d, nq, k = 150_003, 128, 700, 500
    X, Q = fx.unit_rows(rng, N, d), fx.unit_rows(rng, nq, d)
    Q[:5] = X[-5:]                                 # self-matches in the last (partial) stage
    idx = FAISSIndex(embed_dim=d, exact=True)
    idx.build_ivf_index(X, list(range(N)))
    sn, rn = idx.batch_search(Q, k=k)              # 256-query filter (the default)
    monkeypatch.setenv("RIHIP_FILTER_WIDE", "1")
    sw, rw = idx.batch_search(Q, k=k)              # wide filter
    # the same survivors reach the exact re-score; its f32 summation order depends on where a candidate sits in its
    # segment, so scores may differ in the last bit and near-ties may swap places: same SETS, scores to 1e-6
    np.testing.assert_allclose(sw, sn, atol=1e-6, rtol=0)
    for q in range(nq):
        assert set(rw[q]) == set(rn[q]) or np.abs(sn[q, -1] - sn[q, -2]) < 1e-6, q
    assert (rw == rn).mean() > 0.999
    assert (rw[:5, 0] == np.arange(N - 5, N)).all()
    sel = rng.choice(nq, 32, replace=False)
    _check_topk(sw[sel], rw[sel], Q[sel], X, k)
    _lib.check(_lib.lib().rihip_ip_index_set_two_precision(idx.index._h, 0))
    s1, r1 = idx.batch_search(Q, k=k)              # all-f32
    np.testing.assert_allclose(sw, s1, atol=1e-6, rtol=0)
    assert (r1 == rw).mean() > 0.999
    for k2 in (10, 100):                           # other k (other thresholds / segment capacities) on the same index
        _lib.check(_lib.lib().rihip_ip_index_set_two_precision(idx.index._h, 1))
        s2, r2 = idx.batch_search(Q, k=k2)
        _check_topk(s2[sel], r2[sel], Q[sel], X, k2)


def test_two_precision_scratch_reuse_across_batch_sizes():
    """The filter's per-(query, corpus split) segments live in scratch that is reused by later calls with another
    number of splits; splits that get no rows (the split count does not divide the stages) must still publish an empty
    segment (regression: stale counts leaked candidates of an earlier batch into the result)."""
    rng = np.random.RandomState(13)
    N, d, k = 250000, 128, 500
    X = fx.unit_rows(rng, N, d)
    idx = _index(X)
    Qa, Qb = fx.unit_rows(rng, 1500, d), fx.unit_rows(rng, 200, d)
    idx.batch_search(Qa, k=k)                      # 6 query blocks -> 128 splits
    s, r = idx.batch_search(Qb, k=k)               # 1 query block  -> 768 splits, 116 of them empty
    sel = rng.choice(200, 24, replace=False)
    _check_topk(s[sel], r[sel] - 1000, Qb[sel], X, k)
    s2, r2 = idx.batch_search(Qb, k=k)
    np.testing.assert_array_equal(r2, r)


def test_two_precision_dense_survivors_queue_overflow():
    """Small N / large query batch: ~10 % of the 16-score columns hold a survivor, so the per-workgroup LDS queue
    overflows into the out-of-line append path (regression: this shape used to fault)."""
    from recommendit_amd import FAISSIndex
    rng = np.random.RandomState(12)
    N, d, nq, k = 100000, 128, 1500, 500
    X, Q = fx.unit_rows(rng, N, d), fx.unit_rows(rng, nq, d)
    idx = FAISSIndex(embed_dim=d, exact=True)
    idx.build_ivf_index(X, list(range(N)))
    s, r = idx.batch_search(Q, k=k)
    sel = rng.choice(nq, 24, replace=False)
    _check_topk(s[sel], r[sel], Q[sel], X, k)
    assert (np.diff(s, axis=1) <= 0).all() and (r >= 0).all()


def test_ivf_large_lists_thresholded_path():
    """Lists large enough for the sampled-threshold IVF scan: with nprobe == nlist the result must be the exact
    top-k; with nprobe < nlist every returned (score,row) is a true inner product, sorted, k of them, and the
    batch result equals the one-query-at-a-time result (different blocks / tile lists / thresholds)."""
    from recommendit_amd import FAISSIndex
    rng = np.random.RandomState(21)
    N, d, nq, k = 300000, 64, 150, 100
    X, Q = fx.unit_rows(rng, N, d), fx.unit_rows(rng, nq, d)
    ivf = FAISSIndex(embed_dim=d, n_lists=50, n_probe=50)
    ivf.build_ivf_index(X, list(range(N)), kmeans_iters=5)
    sc, rows = ivf.batch_search(Q, k=k)
    _check_topk(sc, rows, Q, X, k)
    ivf.set_n_probe(5)
    sc5, rows5 = ivf.batch_search(Q, k=k)
    assert (rows5 >= 0).all() and (np.diff(sc5, axis=1) <= 0).all()
    S = Q.astype(np.float64) @ X.astype(np.float64).T
    np.testing.assert_allclose(sc5, np.take_along_axis(S, rows5, 1), atol=TOL)
    for q in (0, 7, 149):
        s1, r1 = ivf.search(Q[q], k=k)
        assert list(r1) == list(rows5[q])
    ref_sc, ref_rows = R.topk_ip_exact(Q, X, k)
    recall = np.mean([len(set(rows5[q]) & set(ref_rows[q])) / k for q in range(nq)])
    assert recall > 0.3, recall


@pytest.mark.parametrize("kind", ["flat_small", "flat_two_precision", "ivf"])
def test_in_search_item_id_mapping_equals_the_separate_pass(kind):
    """rihip_ip_index_set_id_map: the search's last kernel writes item ids (faiss_index.py:123,148-152) -- same result as
    mapping the returned rows afterwards, -1 padding untouched, and switching the map off restores row numbers."""
    from recommendit_amd import _lib as L
    rng = np.random.default_rng(5)
    N, d, nq, k = (3000, 64, 40, 500) if kind != "flat_two_precision" else (70000, 128, 50, 500)
    X = rng.standard_normal((N, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    ids = rng.permutation(10 * N)[:N].astype(np.int64) + 1
    from recommendit_amd import FAISSIndex
    idx = FAISSIndex(embed_dim=d, exact=(kind != "ivf"), n_lists=30, n_probe=2)
    idx.build_ivf_index(X, ids)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    qd = torch.from_numpy(Q).to(L.device())
    s_rows, rows = idx._search_device(qd, k)
    s_ids, mapped = idx._search_device(qd, k, item_ids=True)
    rows, mapped = rows.cpu().numpy(), mapped.cpu().numpy()
    assert np.array_equal(s_rows.cpu().numpy(), s_ids.cpu().numpy())
    want = np.where(rows >= 0, ids[np.maximum(rows, 0)], -1)
    assert np.array_equal(mapped, want)
    if kind == "ivf":
        assert (rows < 0).any()                                   # short results exist: the padding stayed -1
    _, again = idx._search_device(qd, k)                          # map switched off again
    assert np.array_equal(again.cpu().numpy(), rows)
