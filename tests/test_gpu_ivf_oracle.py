"""GPU parity of the IVF-Flat (inner product) path against oracle/retrieval_np.py -- the only retrieval mode the
reference has (src/models/faiss_index.py:68-74 build: 100 lists, nprobe 10; :113/:145 search).

faiss is not importable here, so this is *parity unpinned vs faiss, pinned to the oracle*: the oracle restates the
IndexIVFFlat semantics (IndexFlatIP quantizer = arg-max inner product; coarse top-nprobe lists; exact scan of the probed
lists; -1 padding) and the HIP kernels must reproduce it from the SAME centroids and list membership:
  (i)   ivf_assign_mfma_kernel        vs ivf_assign            (bit-exact arg-max except float near-ties)
  (ii)  rihip_ip_index_search, nprobe < nlist  vs ivf_search   (identical rows / scores / -1 padding) at BASELINE cfg5
        (N=1M, d=128, 100 lists, nprobe 10, k=500) and at the ML-1M shape (N=3883, d=64, 99 lists: k' < 500)
  (iii) Lloyd iterations                vs kmeans_ip            from the same initial centroids
Float tolerance: exact-f32 fmaf chains vs float64, 2e-6 on unit vectors; a comparison is skipped only where the
oracle itself shows a margin below 4x that tolerance (coarse probe boundary, k-th score boundary, arg-max margin)."""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import retrieval_np as R

pytestmark = pytest.mark.gpu
TOL = 2e-6


def _build(X, nlist, nprobe, **kw):
    from recommendit_amd import FAISSIndex
    idx = FAISSIndex(embed_dim=X.shape[1], n_lists=nlist, n_probe=nprobe)
    idx.build_ivf_index(X, list(range(X.shape[0])), **kw)
    return idx


def _clustered(rng, N, d, n_centers, spread=0.35):
    """unit rows around n_centers directions: what item-tower outputs look like (lists of uneven size)"""
    centers = fx.unit_rows(rng, n_centers, d)
    w = rng.dirichlet(np.full(n_centers, 0.7))
    which = rng.choice(n_centers, N, p=w)
    X = centers[which] + spread * rng.randn(N, d).astype(np.float32) / np.sqrt(d)
    return R.normalize_rows(X)


def _compare_search(idx, Q, X, nprobe, k, min_checked=0.9):
    C, a = idx.centroids(), idx.list_assignment()
    sc, rows = idx.batch_search(Q, k=k)
    kk = min(k, X.shape[0])
    o_s, o_r, probe, coarse = R.ivf_search(Q, X, C, a, nprobe, kk, return_probe=True)
    srt = -np.sort(-coarse, axis=1)
    checked = 0
    for q in range(Q.shape[0]):
        if nprobe < C.shape[0] and srt[q, nprobe - 1] - srt[q, nprobe] < 4 * TOL:
            continue                                   # coarse boundary is a float near-tie: either list set is right
        checked += 1
        n_ok = int((o_r[q] >= 0).sum())
        assert int((rows[q] >= 0).sum()) == n_ok, (q, n_ok)                 # same number of results, same -1 padding
        assert (rows[q, n_ok:] == -1).all() and np.isneginf(sc[q, n_ok:]).all()
        np.testing.assert_allclose(sc[q, :n_ok], o_s[q, :n_ok], atol=TOL, rtol=0)
        if (rows[q, :n_ok] == o_r[q, :n_ok]).all():
            continue
        # order/membership may differ only among float near-ties
        diff = np.nonzero(rows[q, :n_ok] != o_r[q, :n_ok])[0]
        for i in diff:                                   # position i ties with a neighbour (f32 vs f64 rounding)
            gaps = [abs(float(o_s[q, i]) - float(o_s[q, j])) for j in (i - 1, i + 1) if 0 <= j < n_ok]
            assert min(gaps) < 4 * TOL or i == n_ok - 1, (q, i, gaps)
        extra = set(rows[q, :n_ok].tolist()) ^ set(o_r[q, :n_ok].tolist())
        if extra:                                       # a swap across the k-th boundary: scores equal within TOL
            true = X[sorted(extra)].astype(np.float64) @ Q[q].astype(np.float64)
            assert np.abs(true - o_s[q, n_ok - 1]).max() < 4 * TOL, (q, extra)
    assert checked >= min_checked * Q.shape[0], checked
    return sc, rows, o_s, o_r


def test_assign_kernel_vs_oracle():
    rng = np.random.RandomState(31)
    for d, nlist, N in ((128, 100, 60000), (64, 99, 3883), (32, 257, 20000)):
        X = fx.unit_rows(rng, N, d)
        idx = _build(X, nlist, 1, kmeans_iters=2)
        C = idx.centroids()
        assert C.shape == (nlist, d)
        Y = fx.unit_rows(rng, 5000 + 17, d)            # rows that are not in the index, ragged count
        got = idx.assign_lists(torch.from_numpy(Y).cuda()).cpu().numpy()
        S = Y.astype(np.float64) @ C.astype(np.float64).T
        ref = R.ivf_assign(Y, C)
        srt = -np.sort(-S, axis=1)
        clear = srt[:, 0] - srt[:, 1] > 4 * TOL
        assert clear.mean() > 0.99
        np.testing.assert_array_equal(got[clear], ref[clear])
        # near-ties: the chosen list is one of the (numerically) best
        np.testing.assert_allclose(S[np.arange(len(Y)), got], srt[:, 0], atol=4 * TOL)
        # the stored membership IS the assignment of the stored rows to the final centroids
        a = idx.list_assignment()
        Sx = X.astype(np.float64) @ C.astype(np.float64).T
        sx = -np.sort(-Sx, axis=1)
        cl = sx[:, 0] - sx[:, 1] > 4 * TOL
        np.testing.assert_array_equal(a[cl], R.ivf_assign(X, C)[cl])


def test_assign_ties_pick_lowest_list():
    """integer-valued rows/centroids: products are exact in f32, ties must go to the lowest list id (np.argmax)"""
    from recommendit_amd import FAISSIndex
    rng = np.random.RandomState(32)
    d, nlist, N = 32, 40, 4096
    C = rng.randint(-2, 3, size=(nlist, d)).astype(np.float32)
    C[7] = C[3]; C[35] = C[3]                                        # duplicate centroids -> guaranteed ties
    X = rng.randint(-2, 3, size=(N, d)).astype(np.float32)
    idx = FAISSIndex(embed_dim=d, n_lists=nlist, n_probe=4)
    idx.build_from_device(torch.from_numpy(X).cuda(), np.arange(N), kmeans_iters=0, init_centroids=C)
    np.testing.assert_array_equal(idx.list_assignment(), R.ivf_assign(X, C))
    np.testing.assert_array_equal(idx.centroids(), C)
    np.testing.assert_array_equal(idx.reconstruct(), X)


def test_lloyd_iterations_vs_oracle():
    rng = np.random.RandomState(33)
    from recommendit_amd import FAISSIndex
    for d, nlist, N, iters in ((64, 50, 30000, 1), (128, 100, 50000, 3)):
        X = _clustered(rng, N, d, 30)
        init = X[rng.choice(N, nlist, replace=False)]
        idx = FAISSIndex(embed_dim=d, n_lists=nlist, n_probe=5)
        idx.build_from_device(torch.from_numpy(X).cuda(), np.arange(N), kmeans_iters=iters, init_centroids=init)
        C = idx.centroids()
        Co = R.kmeans_ip(X, nlist, n_iter=iters, init=init)
        # f32 partial sums in a fixed order vs float64 means; a near-tie row that flips list moves a mean by ~1/|list|
        close = np.abs(C - Co).max(axis=1)
        assert (close < 1e-5).mean() > 0.9 and close.max() < 5e-3, (close.max(), (close < 1e-5).mean())
        a = idx.list_assignment()
        ao = R.ivf_assign(X, Co)
        assert (a == ao).mean() > 0.999
        # bitwise reproducible build
        idx2 = FAISSIndex(embed_dim=d, n_lists=nlist, n_probe=5)
        idx2.build_from_device(torch.from_numpy(X).cuda(), np.arange(N), kmeans_iters=iters, init_centroids=init)
        np.testing.assert_array_equal(idx2.centroids(), C)
        np.testing.assert_array_equal(idx2.list_assignment(), a)


def test_set_ivf_roundtrip_and_more_lists_than_lds_could_hold(tmp_path):
    """inject a partition (what reading a FAISS file does), 512 lists at d=128 (round 1 was limited to 127), save/load"""
    from recommendit_amd import FAISSIndex
    rng = np.random.RandomState(34)
    N, d, nlist = 40000, 128, 512
    X = fx.unit_rows(rng, N, d)
    C = fx.unit_rows(rng, nlist, d)
    a = R.ivf_assign(X, C).astype(np.int32)
    idx = FAISSIndex(embed_dim=d, n_lists=nlist, n_probe=16)
    idx.build_from_device(torch.from_numpy(X).cuda(), np.arange(N), centroids=C, assign=a)
    np.testing.assert_array_equal(idx.list_assignment(), a)
    np.testing.assert_array_equal(idx.centroids(), C)
    Q = fx.unit_rows(rng, 40, d)
    sc, rows, _, _ = _compare_search(idx, Q, X, 16, 100)
    idx.save(str(tmp_path / "i.idx"))
    idx2 = FAISSIndex.load(str(tmp_path / "i.idx"))
    sc2, rows2 = idx2.batch_search(Q, k=100)
    np.testing.assert_array_equal(rows2, rows)
    np.testing.assert_array_equal(sc2, sc)
    # k-means itself at a list count that did not fit LDS before
    idx3 = FAISSIndex(embed_dim=d, n_lists=nlist, n_probe=16)
    idx3.build_ivf_index(X, list(range(N)), kmeans_iters=2)
    assert idx3.centroids().shape == (nlist, d) and np.bincount(idx3.list_assignment(), minlength=nlist).sum() == N
    _compare_search(idx3, Q, X, 16, 100)


def test_ivf_search_ml1m_shape_short_results():
    """ML-1M catalogue: N=3883 < 3900 -> IndexBuilder shrinks n_lists to N//39 = 99 (src/training/build_index.py:119-126);
    nprobe 10 of 99 lists holds ~390 < 500 vectors, so the reference returns fewer than k (faiss_index.py:119-121)."""
    rng = np.random.RandomState(35)
    N, d, nlist, nprobe, k = 3883, 64, 99, 10, 500
    X = _clustered(rng, N, d, 18)
    idx = _build(X, nlist, nprobe)
    Q = np.concatenate([fx.unit_rows(rng, 100, d), X[:28]])
    sc, rows, o_s, o_r = _compare_search(idx, Q, X, nprobe, k)
    short = (rows < 0).any(axis=1)
    assert short.any()                                          # the short-result case really occurs
    # single-query entry: invalid slots are dropped (faiss_index.py:119-121)
    for q in (0, 5, 127):
        s1, ids1 = idx.search(Q[q], k=k)
        n_ok = int((o_r[q] >= 0).sum())
        assert len(ids1) == n_ok and list(ids1) == list(rows[q, :n_ok])


@pytest.mark.parametrize("clustered", [False, True])
def test_ivf_search_cfg5_1m_100_lists_nprobe10_k500(clustered):
    """BASELINE.json configs[4]: N=1M, d=128, IVF-IP 100 lists, nprobe 10, 500 candidates."""
    rng = np.random.RandomState(36 + int(clustered))
    N, d, nlist, nprobe, k = 1_000_000, 128, 100, 10, 500
    X = _clustered(rng, N, d, 64) if clustered else fx.unit_rows(rng, N, d)
    idx = _build(X, nlist, nprobe, kmeans_iters=4)
    Q = np.concatenate([fx.unit_rows(rng, 90, d), X[rng.choice(N, 38, replace=False)]])
    sc, rows, o_s, o_r = _compare_search(idx, Q, X, nprobe, k)
    assert (rows >= 0).all()
    # one query at a time == batch (different query blocks / tile lists / thresholds)
    for q in (0, 64, 127):
        s1, r1 = idx.search(Q[q], k=k)
        assert list(r1) == list(rows[q])


def test_ivf_underfill_not_masked_by_list_padding():
    """padding rows (score 0) must not count as candidates.  An anti-correlated query (all scores <= 0) over
    lists whose lengths sit just above a multiple of 64 (=> up to 63 zero rows per list) must still return the true
    top-k of the probed lists."""
    from recommendit_amd import FAISSIndex
    rng = np.random.RandomState(38)
    d, nlist, nprobe, k = 64, 20, 10, 500
    m = fx.unit_rows(rng, 1, d)
    C = R.normalize_rows(m + fx.unit_rows(rng, nlist, d))     # every centroid inside a cone around m
    per = 64 * 60 + 1                                          # 3841 rows per list -> 63 padding rows each; large
    # enough (10 x 3904 probed rows) for the sampled-threshold scan, where a zero-score padding row passes thr <= 0
    X = np.concatenate([R.normalize_rows(C[c] + 0.3 * rng.randn(per, d).astype(np.float32) / np.sqrt(d))
                        for c in range(nlist)])
    a = np.repeat(np.arange(nlist), per).astype(np.int32)
    perm = rng.permutation(len(X)); X, a = X[perm], a[perm]
    idx = FAISSIndex(embed_dim=d, n_lists=nlist, n_probe=nprobe)
    idx.build_from_device(torch.from_numpy(X).cuda(), np.arange(len(X)), centroids=C, assign=a)
    Q = R.normalize_rows(-m + 0.2 * fx.unit_rows(rng, 8, d))   # negative scores against every list
    sc, rows, o_s, o_r = _compare_search(idx, Q, X, nprobe, k, min_checked=0.5)
    assert (rows >= 0).all() and (sc < 0).all()


def test_faiss_file_interop(tmp_path):
    """§8f-3: (a) an index trained here, written as a FAISS IndexIVFFlat file and read back, searches identically;
    (b) a 'foreign' IndexIVFFlat file (centroids / lists not produced by this library's trainer) is loaded with ITS
    partition and searched like the oracle; (c) flat indexes likewise.  Byte layout restated from faiss 1.7.x (parity
    unpinned: faiss is not importable here, the reference ships no index file)."""
    import pickle
    from recommendit_amd import FAISSIndex, faiss_io
    rng = np.random.RandomState(40)
    N, d, nlist, nprobe, k = 30000, 64, 40, 6, 100
    X = _clustered(rng, N, d, 20)
    Q = fx.unit_rows(rng, 60, d)
    ids = list(range(500, 500 + N))
    idx = FAISSIndex(embed_dim=d, n_lists=nlist, n_probe=nprobe)
    idx.build_ivf_index(X, ids)
    s0, r0 = idx.batch_search(Q, k=k)
    idx.save(str(tmp_path / "a.index"), format="faiss")
    assert faiss_io.sniff(str(tmp_path / "a.index")) == "faiss"
    back = FAISSIndex.load(str(tmp_path / "a.index"))
    s1, r1 = back.batch_search(Q, k=k)
    np.testing.assert_array_equal(r1, r0)
    np.testing.assert_array_equal(s1, s0)
    np.testing.assert_array_equal(back.list_assignment(), idx.list_assignment())
    # (b) foreign file: NumPy k-means partition written in the FAISS layout + the reference's .meta.pkl sidecar
    C = R.kmeans_ip(X, nlist, n_iter=3, seed=7)
    a = R.ivf_assign(X, C)
    Xn = R.normalize_rows(X)
    faiss_io.write_ivf_flat(str(tmp_path / "b.index"), Xn, C, a, nprobe=3)
    with open(tmp_path / "b.meta.pkl", "wb") as f:
        pickle.dump({"item_ids": np.asarray(ids), "item_id_to_faiss_idx": {i: j for j, i in enumerate(ids)},
                     "embed_dim": d, "n_lists": nlist, "n_probe": nprobe}, f)
    fb = FAISSIndex.load(str(tmp_path / "b.index"))
    assert fb.index.is_ivf and fb.index.ntotal == N and fb.n_probe == nprobe       # sidecar's nprobe wins (faiss_index.py:197)
    np.testing.assert_array_equal(fb.list_assignment(), a)
    np.testing.assert_array_equal(fb.centroids(), C)
    sc, ids_got = fb.batch_search(Q, k=k)
    o_s, o_r = R.ivf_search(R.normalize_rows(Q), Xn, C, a, nprobe, k)
    np.testing.assert_allclose(sc, o_s, atol=TOL, rtol=0)
    assert (ids_got == np.asarray(ids)[o_r]).mean() > 0.995                         # near-tie swaps only
    # (c) flat
    ex = FAISSIndex(embed_dim=d, exact=True)
    ex.build_ivf_index(X, ids)
    ex.save(str(tmp_path / "c.index"), format="faiss")
    ex2 = FAISSIndex.load(str(tmp_path / "c.index"))
    assert ex2.exact and not ex2.index.is_ivf
    np.testing.assert_array_equal(ex2.batch_search(Q, k=k)[1], ex.batch_search(Q, k=k)[1])


def test_faiss_file_load_guards(tmp_path):
    """IndexIVFFlat files the HIP index cannot hold as lists -- more lists than vectors -> served as a flat
    (exact) index; more than 2048 lists -> a FaissFormatError, not a bare assert; rank_topk serves k up to 16384."""
    import pickle
    from recommendit_amd import FAISSIndex, faiss_io
    rng = np.random.RandomState(41)
    d = 32

    def write(name, n, nlist):
        X = fx.unit_rows(rng, n, d)
        C = fx.unit_rows(rng, nlist, d)
        a = R.ivf_assign(X, C)
        faiss_io.write_ivf_flat(str(tmp_path / f"{name}.index"), X, C, a, nprobe=4)
        ids = np.arange(10, 10 + n)
        with open(tmp_path / f"{name}.meta.pkl", "wb") as f:
            pickle.dump({"item_ids": ids, "item_id_to_faiss_idx": {int(i): j for j, i in enumerate(ids)},
                         "embed_dim": d, "n_lists": nlist, "n_probe": 4}, f)
        return X, ids

    X, ids = write("tiny", 50, 80)                       # nlist > ntotal
    idx = FAISSIndex.load(str(tmp_path / "tiny.index"))
    assert idx.exact and idx.index.ntotal == 50
    Q = fx.unit_rows(rng, 5, d)
    sc, got = idx.batch_search(Q, k=10)
    o_s, o_r = R.topk_ip_exact(Q, X, 10)
    np.testing.assert_allclose(sc, o_s, atol=TOL)
    assert (got == ids[o_r]).mean() > 0.95
    write("wide", 5000, 2500)                            # nlist > 2048
    with pytest.raises(faiss_io.FaissFormatError, match="2048"):
        FAISSIndex.load(str(tmp_path / "wide.index"))


def test_rank_topk_large_candidate_sets_and_nan_scores():
    """rihip_rank_topk (DataFrame.nlargest at recommender.py:346) up to the index's K_MAX = 16384 candidates; NaN ranker
    scores never outrank a number, padded candidates come last"""
    from recommendit_amd import _lib as L
    lib, dev = L.lib(), L.device()
    rng = np.random.RandomState(5)
    for nq, kc, k in ((3, 16384, 50), (2, 9000, 9000), (4, 500, 20)):
        s = rng.randn(nq, kc)
        s[:, 7] = np.nan; s[0, 11] = np.inf; s[1, 13] = -np.inf
        cand = np.arange(nq * kc, dtype=np.int64).reshape(nq, kc) + 5
        cand[:, kc - 3:] = -1
        rs = rng.rand(nq, kc).astype(np.float32)
        sd, cd, rd = (torch.from_numpy(x).to(dev) for x in (s, cand, rs))
        ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        top = torch.empty((nq, k), dtype=torch.float64, device=dev)
        trs = torch.empty((nq, k), dtype=torch.float32, device=dev)
        L.check(lib.rihip_rank_topk(sd.data_ptr(), cd.data_ptr(), rd.data_ptr(), nq, kc, k, ids.data_ptr(), top.data_ptr(),
                                    trs.data_ptr(), L.stream_ptr()), "rank_topk")
        ids, top = ids.cpu().numpy(), top.cpu().numpy()
        for q in range(nq):
            key = np.where(cand[q] < 0, -np.inf, s[q])
            cls = np.where(cand[q] < 0, 2, np.where(np.isnan(s[q]), 1, 0))       # numbers, then NaN, then padding
            order = np.lexsort((np.arange(kc), -np.nan_to_num(key, nan=0.0, posinf=1e308, neginf=-1e308), cls))[:k]
            np.testing.assert_array_equal(ids[q], cand[q][order])
            if k < kc - 4:
                assert not np.isnan(top[q]).any()


def test_deferred_exactness_check_equals_the_synchronous_search():
    """Serving chains defer the thresholded IVF pass's exactness check to their end (`set_deferred_check` /
    `finish_search`): the deferred search + finish must give what the synchronous search gives, including for queries
    that take the exact re-do (here: 6 000 identical rows tie at the threshold and overflow the candidate lists)."""
    from recommendit_amd import FAISSIndex
    rng = np.random.RandomState(5)
    N, d, nq, k = 300_000, 128, 600, 500
    X = fx.unit_rows(rng, N, d)
    X[:6000] = X[0]
    Q = fx.unit_rows(rng, nq, d)
    Q[:40] = X[0] + 0.01 * rng.randn(40, d).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    idx = FAISSIndex(embed_dim=d, n_lists=100, n_probe=10)
    idx.build_from_device(torch.from_numpy(X).cuda(), np.arange(N))
    q = torch.from_numpy(Q).cuda()
    s0, r0 = idx.batch_search_device(q, k=k, normalized=True)            # synchronous check + re-do inside
    idx.set_deferred_check(True)
    s1, r1 = idx.batch_search_device(q, k=k, normalized=True)
    assert idx.search_pending()
    n = idx.finish_search()
    idx.set_deferred_check(False)
    assert n >= 40, n                                                     # the tied queries were re-done
    assert not idx.search_pending()
    torch.testing.assert_close(r1, r0, rtol=0, atol=0)
    torch.testing.assert_close(s1, s0, rtol=0, atol=0)
    # a batch of random queries (few or no re-dos): same results
    qb = torch.from_numpy(fx.unit_rows(rng, nq, d)).cuda()
    s2, r2 = idx.batch_search_device(qb, k=k, normalized=True)
    idx.set_deferred_check(True)
    s3, r3 = idx.batch_search_device(qb, k=k, normalized=True)
    assert idx.finish_search() < 40
    idx.set_deferred_check(False)
    torch.testing.assert_close(r3, r2, rtol=0, atol=0)
    torch.testing.assert_close(s3, s2, rtol=0, atol=0)
