"""Randomised shapes for the in-batch passes (stored-G and two-sweep forms, f32 MFMA and bf16x6) against the oracle:
ragged owner/swept counts, rectangular rank-local shapes with an offset diagonal, swept-range splits."""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import two_tower_np as O

pytestmark = pytest.mark.gpu


def _run(Bl, G, off, d, prec, store_g):
    from recommendit_amd import _lib as L
    lib, dev, st = L.lib(), L.device(), L.stream_ptr()
    rng = np.random.RandomState(Bl * 7 + G)
    U, I = fx.unit_rows(rng, Bl, d), fx.unit_rows(rng, G, d)
    Ud, Id = torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev)
    f32 = dict(dtype=torch.float32, device=dev)
    pos = torch.empty(Bl, **f32); r = torch.empty(Bl, **f32)
    dU = torch.full((Bl, d), float("nan"), **f32)
    lp = torch.zeros(max(1024, lib.rihip_inbatch_workspace_doubles(Bl)), dtype=torch.float64, device=dev)
    ws = torch.empty(max(lib.rihip_inbatch_workspace_floats(Bl, G, d), lib.rihip_inbatch_workspace_floats(G, Bl, d)), **f32)
    loss = torch.empty((), **f32)
    L.check(lib.rihip_rowdot(Ud.data_ptr(), Id.data_ptr(), Bl, off, d, pos.data_ptr(), st), "rowdot")
    if store_g:
        dI = torch.full((G, d), float("nan"), **f32)
        gm = torch.full((lib.rihip_inbatch_gmat_floats(Bl, G),), float("nan"), **f32)
        L.check(lib.rihip_inbatch_user_pass(Ud.data_ptr(), Bl, off, Id.data_ptr(), G, 0, d, pos.data_ptr(), G,
                                            dU.data_ptr(), r.data_ptr(), lp.data_ptr(), ws.data_ptr(), gm.data_ptr(),
                                            prec, st), "user_pass")
        L.check(lib.rihip_inbatch_item_pass(gm.data_ptr(), Ud.data_ptr(), Bl, off, G, 0, d, r.data_ptr(), G,
                                            dI.data_ptr(), ws.data_ptr(), prec, st), "item_pass")
    else:
        dI = None
        L.check(lib.rihip_inbatch_sweep(1, Ud.data_ptr(), Bl, off, Id.data_ptr(), G, 0, d, pos.data_ptr(), None, G,
                                        dU.data_ptr(), r.data_ptr(), lp.data_ptr(), ws.data_ptr(), prec, st), "sweep")
    L.check(lib.rihip_sum_partials(lp.data_ptr(), lib.rihip_inbatch_loss_parts(Bl, G), 1.0 / (G * (G - 1.0)),
                                   loss.data_ptr(), st), "sum")
    lo, dUo, dIo = O.in_batch_bpr_loss(U, I, owner_offset=off, n_global=G)
    assert abs(loss.item() - float(lo)) < 3e-6, (loss.item(), float(lo))
    np.testing.assert_allclose(dU.cpu().numpy(), dUo, atol=3e-9, rtol=3e-4)
    if dI is not None:
        np.testing.assert_allclose(dI.cpu().numpy(), dIo, atol=3e-9, rtol=3e-4)


def test_inbatch_random_shapes_vs_oracle():
    rng = np.random.RandomState(2024)
    cases = [(1, 2, 0), (1, 2, 1), (2, 2, 0), (31, 33, 2), (32, 32, 0), (33, 31 + 33, 31), (129, 257, 128)]
    for _ in range(22):
        Bl = int(rng.randint(1, 700))
        G = Bl + int(rng.randint(0, 900))
        cases.append((Bl, G, int(rng.randint(0, G - Bl + 1))))
    for i, (Bl, G, off) in enumerate(cases):
        d = (32, 64, 128)[i % 3]
        prec = (0, 2)[(i // 3) % 2]
        _run(Bl, G, off, d, prec, store_g=(i % 4 != 3))


def test_retrieval_random_shapes_exact():
    """Two-precision search (sample threshold -> bf16 filter -> provable survivor cut -> exact re-score) against a
    plain fp32 top-k on the device, over corpus sizes around the brute-force / sampled-path switch, with clustered
    scores (many near-ties) and duplicated rows."""
    from recommendit_amd import FAISSIndex
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(77)
    for case, (N, d, nq, k) in enumerate([(70000, 32, 33, 10), (131072, 64, 257, 500), (300000, 128, 64, 100),
                                          (90001, 64, 5, 1), (262144, 32, 100, 500)]):
        X = torch.randn(N, d, generator=g)
        if case % 2 == 1:   # clustered: low-rank structure + small noise -> dense near-ties around the k-th score
            X = torch.randn(N, 4, generator=g) @ torch.randn(4, d, generator=g) + 0.01 * X
        X[N // 2:N // 2 + 50] = X[:50]            # exact duplicates
        X = torch.nn.functional.normalize(X, dim=1).contiguous()
        Q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1).contiguous()
        idx = FAISSIndex(embed_dim=d, exact=True)
        idx.build_from_device(X.to(dev), np.arange(N))
        sc, rows = idx.batch_search_device(Q.to(dev), k=k, normalized=True)
        ref = (Q.to(dev) @ X.to(dev).T)
        rs, _ = torch.topk(ref, k, dim=1)
        # scores must match the exact top-k scores (rows may differ only among exactly tied scores)
        np.testing.assert_allclose(sc.cpu().numpy(), rs.cpu().numpy(), atol=2e-6, rtol=0, err_msg=str((N, d, nq, k)))
        got = torch.gather(ref, 1, rows.clamp_min(0))
        np.testing.assert_allclose(got.cpu().numpy(), sc.cpu().numpy(), atol=2e-6, rtol=0)
        assert all(len(set(r.tolist())) == k for r in rows.cpu())
