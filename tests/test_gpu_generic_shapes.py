"""Arbitrary (embed_dim, hidden_dim): the reference builds its towers from any pair (src/models/two_tower.py:27-33, :54-62,
:80-95; EMBEDDING_DIM is env-overridable, src/config.py:13).  Pairs without a tuned template instantiation run the
runtime-shape kernels of csrc/tower_generic.hip -- held here to the SAME tolerances against the NumPy oracle as the tuned
kernels in tests/test_gpu_towers.py (forward 2e-6, gradients rtol 3e-4), including the counter-based dropout mask, the
in-batch loss (zero-padded embedding columns), the fused trainer step and retrieval over a zero-padded index."""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import two_tower_np as O

pytestmark = pytest.mark.gpu

SHAPES = [(48, 96), (96, 256), (16, 32), (256, 256), (144, 80), (128, 64), (64, 256)]


def t(x):
    return torch.from_numpy(np.asarray(x)).to("cuda:0")


def _model(nu, ni, d, H, seed, dropout=0.0):
    from recommendit_amd import TwoTowerModel
    sd = fx.make_state(nu, ni, d, H, seed)
    m = TwoTowerModel(nu, ni, embed_dim=d, hidden_dim=H, dropout=dropout)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return m, sd


def _params(sd, tower):
    return O.TowerParams(sd[f"{tower}.embedding.weight"], sd[f"{tower}.mlp.0.weight"], sd[f"{tower}.mlp.0.bias"],
                         sd[f"{tower}.mlp.3.weight"], sd[f"{tower}.mlp.3.bias"])


def test_shape_queries():
    from recommendit_amd import _lib as L
    lib = L.lib()
    for d, H in SHAPES:
        assert lib.rihip_tower_shape_ok(d, H) == 1 and lib.rihip_tower_supported(d, H) == 0
    assert lib.rihip_tower_shape_ok(64, 128) == 1 and lib.rihip_tower_supported(64, 128) == 1
    for d, H in ((40, 64), (64, 100), (272, 64), (64, 512), (0, 16)):
        assert lib.rihip_tower_shape_ok(d, H) == 0


@pytest.mark.parametrize("d,H", SHAPES)
@pytest.mark.parametrize("B", [1, 33, 64, 1000])
def test_generic_forward_vs_oracle_ragged(d, H, B):
    nu, ni = 80, 120
    m, sd = _model(nu, ni, d, H, seed=d + H)
    m.eval()
    u, p, gp, _, _ = fx.make_batch(nu, ni, B, seed=B)
    with torch.no_grad():
        U = m.user_tower(t(u)).cpu().numpy()
        P = m.item_tower(t(p), t(gp)).cpu().numpy()
        P2 = m.item_tower(t(p), t(gp)).cpu().numpy()
    Uo, _ = O.tower_forward(_params(sd, "user_tower"), u)
    Po, _ = O.tower_forward(_params(sd, "item_tower"), p, gp)
    np.testing.assert_allclose(U, Uo, atol=2e-6, rtol=0)
    np.testing.assert_allclose(P, Po, atol=2e-6, rtol=0)
    np.testing.assert_array_equal(P, P2)                       # bitwise reproducible


@pytest.mark.parametrize("d,H,B", [(48, 96, 200), (96, 256, 129), (16, 32, 70), (256, 256, 64), (144, 80, 5),
                                   (128, 64, 2500)])
def test_generic_backward_vs_oracle_with_dropout_mask(d, H, B):
    nu, ni, p_drop = 60, 90, 0.25
    m, sd = _model(nu, ni, d, H, seed=9, dropout=p_drop)
    m.train()
    u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=B + 3)
    torch.manual_seed(1234)
    seeds = [int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item()) for _ in range(3)]
    torch.manual_seed(1234)
    U = m.user_tower(t(u)); P = m.item_tower(t(p), t(gp)); N = m.item_tower(t(n), t(gn))
    loss = m.bpr_loss(U, P, N)
    loss.backward()
    pu, pi = _params(sd, "user_tower"), _params(sd, "item_tower")
    ku, kp, kn = (O.dropout_keep_mask(s, 0, B, H, p_drop) for s in seeds)
    Uo, cu = O.tower_forward(pu, u, None, ku, p_drop)
    Po, cp = O.tower_forward(pi, p, gp, kp, p_drop)
    No, cn = O.tower_forward(pi, n, gn, kn, p_drop)
    np.testing.assert_allclose(U.detach().cpu().numpy(), Uo, atol=2e-6)
    np.testing.assert_allclose(N.detach().cpu().numpy(), No, atol=2e-6)
    lo, dU, dP, dN = O.bpr_loss(Uo, Po, No)
    assert abs(loss.item() - float(lo)) < 2e-6
    bu = O.tower_backward(pu, cu, dU); bp = O.tower_backward(pi, cp, dP); bn = O.tower_backward(pi, cn, dN)
    tol = dict(atol=3e-7, rtol=3e-4)
    G = {k: v.grad.cpu().numpy() for k, v in m.named_parameters()}
    np.testing.assert_allclose(G["user_tower.embedding.weight"], O.embedding_scatter_add(nu + 1, u, bu[0]), **tol)
    np.testing.assert_allclose(G["item_tower.embedding.weight"],
                               O.embedding_scatter_add(ni + 1, p, bp[0]) + O.embedding_scatter_add(ni + 1, n, bn[0]), **tol)
    for i, key in enumerate(("mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias"), start=1):
        np.testing.assert_allclose(G[f"user_tower.{key}"], bu[i], err_msg=key, **tol)
        np.testing.assert_allclose(G[f"item_tower.{key}"], bp[i] + bn[i], err_msg=key, **tol)


@pytest.mark.parametrize("d", [48, 16, 96, 144, 256])
@pytest.mark.parametrize("B", [33, 300])
def test_generic_inbatch_loss_vs_oracle(d, B):
    """in_batch_bpr_loss (two_tower.py:132-160) at embedding widths without a tuned sweep instantiation"""
    from recommendit_amd.two_tower import inbatch_loss_and_grads
    rng = np.random.RandomState(d + B)
    U, I = fx.unit_rows(rng, B, d), fx.unit_rows(rng, B, d)
    for store_g in (True, False):
        l, dU, dI = inbatch_loss_and_grads(t(U), t(I), store_g=store_g)
        lo, dUo, dIo = O.in_batch_bpr_loss(U, I)
        assert abs(float(l) - float(lo)) < 3e-6
        assert dU.shape == (B, d) and dI.shape == (B, d)
        np.testing.assert_allclose(dU.cpu().numpy(), dUo, atol=3e-9, rtol=3e-4)
        np.testing.assert_allclose(dI.cpu().numpy(), dIo, atol=3e-9, rtol=3e-4)


@pytest.mark.parametrize("mode,opt,d,H", [("sampled", "dense", 48, 96), ("sampled", "sparse", 96, 256),
                                          ("inbatch", "dense", 16, 32), ("inbatch", "sparse", 144, 80)])
def test_generic_trainer_steps_vs_oracle(mode, opt, d, H):
    """fused steps (towers -> loss -> backward -> clip -> Adam) at generic shapes against the oracle's step
    (the comparison of tests/test_gpu_trainer.py::test_fused_step_variants_vs_oracle)"""
    from recommendit_amd.trainer import HipBPRTrainer
    from test_gpu_trainer import _oracle_step
    nu, ni, B = 70, 110, 48
    m, sd = _model(nu, ni, d, H, seed=3)
    m.train()
    tr = HipBPRTrainer(m, B, lr=5e-3, weight_decay=1e-5, loss_mode=mode, table_opt=opt)
    mom = ({k: np.zeros_like(v) for k, v in sd.items()}, {k: np.zeros_like(v) for k, v in sd.items()})
    for step in range(1, 5):
        u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=step * 11, boundary=False)
        if mode == "sampled":
            loss = tr.step(t(u), t(np.concatenate([p, n])), t(np.concatenate([gp, gn])))
        else:
            loss = tr.step(t(u), t(p), t(gp))
        lo = _oracle_step(sd, mom, u, p, gp, n, gn, step, 5e-3, mode, opt == "sparse")
        assert abs(loss.item() - lo) < 1e-5, (step, loss.item(), lo)
    tr.check_errors()
    # an element whose gradient sits at Adam's eps = 1e-8 turns an ulp-level gradient difference into a visible fraction
    # of one lr-sized step (lr = 5e-3): at most 0.1 % of the elements (or one) may do so, everything else agrees to 3e-5
    for k, prm in m.named_parameters():
        got = prm.detach().cpu().numpy()
        np.testing.assert_allclose(got, sd[k], atol=2e-3, rtol=0, err_msg=k)
        assert np.sum(np.abs(got - sd[k]) > 3e-5) <= max(1, int(1e-3 * got.size)), k


@pytest.mark.parametrize("d", [48, 16, 96, 100, 20])
def test_generic_index_dims_vs_oracle(d):
    """FAISSIndex at embedding widths outside {32, 64, 128} (any width up to 128: rows are zero-padded to the kernel width
    inside the handle): exact search and IVF against the retrieval oracle, state read back at the caller's width"""
    from oracle import retrieval_np as R
    from recommendit_amd import FAISSIndex
    rng = np.random.RandomState(d)
    N, k = 5000, 50
    X = fx.unit_rows(rng, N, d)
    Q = fx.unit_rows(rng, 40, d)
    ids = np.arange(100, 100 + N)
    ex = FAISSIndex(embed_dim=d, exact=True)
    ex.build_ivf_index(X, ids)
    sc, got = ex.batch_search(Q, k=k)
    o_s, o_r = R.topk_ip_exact(R.normalize_rows(Q), R.normalize_rows(X), k)
    np.testing.assert_allclose(sc, o_s, atol=2e-6, rtol=0)
    assert (got == ids[o_r]).mean() > 0.995
    s1, i1 = ex.search(Q[0], k=k)
    np.testing.assert_allclose(s1, o_s[0], atol=2e-6)
    ivf = FAISSIndex(embed_dim=d, n_lists=16, n_probe=4)
    ivf.build_ivf_index(X, ids)
    sc2, got2 = ivf.batch_search(Q, k=k)
    o_s2, o_r2 = R.ivf_search(R.normalize_rows(Q), R.normalize_rows(X), ivf.centroids(), ivf.list_assignment(), 4, k)
    assert ivf.centroids().shape == (16, d)
    np.testing.assert_allclose(sc2, o_s2, atol=2e-6, rtol=0)
    assert (got2 == np.where(o_r2 >= 0, ids[np.maximum(o_r2, 0)], -1)).mean() > 0.99
    assert ivf.stats()["embed_dim"] == d and ivf.reconstruct().shape == (N, d)
    np.testing.assert_allclose(ivf.reconstruct(), R.normalize_rows(X), atol=1e-7)


def test_generic_index_save_load_and_limit(tmp_path):
    from recommendit_amd import FAISSIndex, _lib as L
    import ctypes as C
    rng = np.random.RandomState(1)
    d, N = 48, 3000
    X = fx.unit_rows(rng, N, d); Q = fx.unit_rows(rng, 9, d)
    ids = list(range(7, 7 + N))
    for fmt in ("native", "faiss"):
        idx = FAISSIndex(embed_dim=d, n_lists=12, n_probe=5)
        idx.build_ivf_index(X, ids)
        s0, i0 = idx.batch_search(Q, k=30)
        idx.save(str(tmp_path / f"{fmt}.index"), **({} if fmt == "native" else {"format": "faiss"}))
        back = FAISSIndex.load(str(tmp_path / f"{fmt}.index"))
        assert back.embed_dim == d and back.centroids().shape == (12, d)
        s1, i1 = back.batch_search(Q, k=30)
        np.testing.assert_array_equal(i1, i0)
        np.testing.assert_array_equal(s1, s0)
    h = C.c_void_p()
    assert L.lib().rihip_ip_index_create(144, C.byref(h)) != 0          # the index kernels stop at 128 columns
    assert b"1..128" in L.lib().rihip_last_error()
