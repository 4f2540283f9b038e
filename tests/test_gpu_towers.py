"""GPU parity: HIP towers / losses / backward against the golden vectors captured from the
reference (tests/golden, oracle/make_golden.py) and against the NumPy oracle.

Tolerances (fp32 path, exact-f32 MFMA fmaf chains vs torch-CPU MKL summation order):
  tower outputs (unit-norm rows)  atol 2e-6
  gradients                       rtol 2e-4 + small atol
"""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import two_tower_np as O

pytestmark = pytest.mark.gpu


def _model(nu, ni, d, H, seed, dropout=0.0):
    from recommendit_amd import TwoTowerModel
    sd = fx.make_state(nu, ni, d, H, seed)
    m = TwoTowerModel(nu, ni, embed_dim=d, hidden_dim=H, dropout=dropout)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return m, sd


def _params(sd, tower):
    return O.TowerParams(sd[f"{tower}.embedding.weight"], sd[f"{tower}.mlp.0.weight"], sd[f"{tower}.mlp.0.bias"],
                         sd[f"{tower}.mlp.3.weight"], sd[f"{tower}.mlp.3.bias"])


def dev():
    return torch.device("cuda:0")


def t(x):
    return torch.from_numpy(np.asarray(x)).to(dev())


@pytest.fixture(params=["auto", "rows32", "coop"])
def fwd_kernel(request, monkeypatch):
    """Every tower-kernel family on every shape: "auto" = the library's choice (64-row LDS tiles below ~48k rows),
    "rows32" = the wave-per-32-rows forward and the two-kernel backward (tower2.hip) forced at any size, "coop" = the
    same forward with the one-kernel backward (tower3.hip: data + weight gradients per tile, d = hidden = 128; other
    shapes fall back to the library's choice)."""
    if request.param == "rows32":
        monkeypatch.setenv("RIHIP_TOWER_FWD", "3")
        monkeypatch.setenv("RIHIP_TOWER_BWD", "3")   # two-kernel backward (d = hidden = 128), any size
    elif request.param == "coop":
        monkeypatch.setenv("RIHIP_TOWER_FWD", "3")
        monkeypatch.setenv("RIHIP_TOWER_BWD", "4")   # one-kernel backward (d = hidden = 128), any size
    else:
        monkeypatch.delenv("RIHIP_TOWER_FWD", raising=False)
        monkeypatch.delenv("RIHIP_TOWER_BWD", raising=False)
    return request.param


def test_library_targets_this_gpu():
    import ctypes
    from recommendit_amd import _lib
    buf = ctypes.create_string_buffer(64)
    _lib.check(_lib.lib().rihip_device_arch(buf, 64))
    assert buf.value.decode().startswith("gfx950"), buf.value


@pytest.mark.parametrize("tag", ["small", "ml1m", "d128"])
def test_g1_tower_forward_golden(golden_dir, tag, fwd_kernel):
    g = np.load(golden_dir / "g1_tower_forward.npz")
    nu, ni, d, H, seed = (int(x) for x in g[f"{tag}_cfg"])
    m, sd = _model(nu, ni, d, H, seed)
    m.eval()
    for B in (1, 16, 256):
        u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=100 + B)
        with torch.no_grad():
            U = m.user_tower(t(u)).cpu().numpy()
            P = m.item_tower(t(p), t(gp)).cpu().numpy()
        np.testing.assert_allclose(U, g[f"{tag}_B{B}_U"], atol=2e-6, rtol=0)
        np.testing.assert_allclose(P, g[f"{tag}_B{B}_P"], atol=2e-6, rtol=0)


@pytest.mark.parametrize("cfg", [(100, 200, 32, 64), (100, 200, 64, 128), (50, 60, 128, 128), (50, 60, 64, 64),
                                 (50, 60, 32, 128)])
@pytest.mark.parametrize("B", [1, 63, 64, 65, 1000])
def test_tower_forward_vs_oracle_ragged(cfg, B, fwd_kernel):
    nu, ni, d, H = cfg
    m, sd = _model(nu, ni, d, H, seed=77)
    m.eval()
    u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=B)
    with torch.no_grad():
        U = m.user_tower(t(u)).cpu().numpy()
        P = m.item_tower(t(p), t(gp)).cpu().numpy()
    Uo, _ = O.tower_forward(_params(sd, "user_tower"), u)
    Po, _ = O.tower_forward(_params(sd, "item_tower"), p, gp)
    np.testing.assert_allclose(U, Uo, atol=2e-6, rtol=0)
    np.testing.assert_allclose(P, Po, atol=2e-6, rtol=0)


def test_tower_forward_bitwise_reproducible_and_cpu_inputs():
    m, sd = _model(100, 200, 64, 128, seed=5)
    m.eval()
    u, p, gp, _, _ = fx.make_batch(100, 200, 300, seed=1)
    with torch.no_grad():
        a = m.item_tower(t(p), t(gp))
        b = m.item_tower(t(p), t(gp))
        c = m.item_tower(torch.from_numpy(p), torch.from_numpy(gp))  # CPU inputs -> CPU output
    assert torch.equal(a, b)
    assert c.device.type == "cpu" and torch.equal(a.cpu(), c)
    with pytest.raises(IndexError):
        m.user_tower(torch.tensor([1, 101]))


@pytest.mark.parametrize("tag", ["small", "mid"])
def test_g2_bpr_grads_golden(golden_dir, tag, fwd_kernel):
    g = np.load(golden_dir / "g2_bpr_grads.npz")
    nu, ni, d, H, seed, B = (int(x) for x in g[f"{tag}_cfg"])
    m, sd = _model(nu, ni, d, H, seed)
    m.train()
    u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=200 + B)
    U = m.user_tower(t(u)); P = m.item_tower(t(p), t(gp)); N = m.item_tower(t(n), t(gn))
    U.retain_grad(); P.retain_grad(); N.retain_grad()
    loss = m.bpr_loss(U, P, N)
    loss.backward()
    assert abs(loss.item() - float(g[f"{tag}_loss"])) < 2e-6
    tol = dict(atol=3e-7, rtol=3e-4)
    np.testing.assert_allclose(U.grad.cpu().numpy(), g[f"{tag}_dU"], **tol)
    np.testing.assert_allclose(P.grad.cpu().numpy(), g[f"{tag}_dP"], **tol)
    np.testing.assert_allclose(N.grad.cpu().numpy(), g[f"{tag}_dN"], **tol)
    for k, prm in m.named_parameters():
        np.testing.assert_allclose(prm.grad.cpu().numpy(), g[f"{tag}_grad_{k}"], err_msg=k, **tol)


@pytest.mark.parametrize("cfg", [(100, 200, 32, 64, 200), (100, 200, 64, 128, 129), (50, 60, 128, 128, 70),
                                 (50, 60, 64, 64, 64), (50, 60, 32, 128, 5)])
def test_backward_vs_oracle_with_dropout_mask(cfg, fwd_kernel):
    """train-mode dropout: the kernel's counter-based mask is reproduced bit-for-bit by the oracle."""
    nu, ni, d, H, B = cfg
    p_drop = 0.25
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
    np.testing.assert_allclose(G["user_tower.mlp.0.weight"], bu[1], **tol)
    np.testing.assert_allclose(G["user_tower.mlp.0.bias"], bu[2], **tol)
    np.testing.assert_allclose(G["user_tower.mlp.3.weight"], bu[3], **tol)
    np.testing.assert_allclose(G["user_tower.mlp.3.bias"], bu[4], **tol)
    np.testing.assert_allclose(G["item_tower.mlp.0.weight"], bp[1] + bn[1], **tol)
    np.testing.assert_allclose(G["item_tower.mlp.0.bias"], bp[2] + bn[2], **tol)
    np.testing.assert_allclose(G["item_tower.mlp.3.weight"], bp[3] + bn[3], **tol)
    np.testing.assert_allclose(G["item_tower.mlp.3.bias"], bp[4] + bn[4], **tol)


@pytest.mark.parametrize("B", [2, 16, 256, 96])
def test_g3_inbatch_golden(golden_dir, B):
    from recommendit_amd import TwoTowerModel
    g = np.load(golden_dir / "g3_inbatch.npz")
    m = TwoTowerModel(4, 4, 32, 64)
    U = t(g[f"B{B}_U"]).clone().requires_grad_(True)
    I = t(g[f"B{B}_I"]).clone().requires_grad_(True)
    loss = m.in_batch_bpr_loss(U, I)
    loss.backward()
    assert abs(loss.item() - float(g[f"B{B}_loss"])) < 3e-6
    np.testing.assert_allclose(U.grad.cpu().numpy(), g[f"B{B}_dU"], atol=3e-8, rtol=3e-4)
    np.testing.assert_allclose(I.grad.cpu().numpy(), g[f"B{B}_dI"], atol=3e-8, rtol=3e-4)


@pytest.mark.parametrize("store_g", [True, False])
@pytest.mark.parametrize("B", [2, 16, 256, 96])
def test_g3_inbatch_golden_bf16x6(golden_dir, B, store_g):
    """the reference's own in_batch_bpr_loss outputs (G3) against the split-bf16 mode (precision=2), SAME tolerances"""
    from recommendit_amd.two_tower import inbatch_loss_and_grads
    g = np.load(golden_dir / "g3_inbatch.npz")
    loss, dU, dI = inbatch_loss_and_grads(t(g[f"B{B}_U"]), t(g[f"B{B}_I"]), precision=2, store_g=store_g)
    assert abs(loss.item() - float(g[f"B{B}_loss"])) < 3e-6
    np.testing.assert_allclose(dU.cpu().numpy(), g[f"B{B}_dU"], atol=3e-8, rtol=3e-4)
    np.testing.assert_allclose(dI.cpu().numpy(), g[f"B{B}_dI"], atol=3e-8, rtol=3e-4)


@pytest.mark.parametrize("prec", [0, 2])   # 2 = bf16x6: fp32-level accuracy, held to the SAME tolerances as f32 MFMA
@pytest.mark.parametrize("B,d", [(33, 32), (130, 64), (500, 128), (1024, 64)])
def test_inbatch_vs_oracle_ragged_and_reproducible(B, d, prec):
    from recommendit_amd.two_tower import inbatch_loss_and_grads
    rng = np.random.RandomState(B)
    U, I = fx.unit_rows(rng, B, d), fx.unit_rows(rng, B, d)
    l1, dU1, dI1 = inbatch_loss_and_grads(t(U), t(I), precision=prec)
    l2, dU2, dI2 = inbatch_loss_and_grads(t(U), t(I), precision=prec)
    assert torch.equal(dU1, dU2) and torch.equal(dI1, dI2) and torch.equal(l1, l2)
    lo, dUo, dIo = O.in_batch_bpr_loss(U, I)
    assert abs(l1.item() - float(lo)) < 3e-6
    np.testing.assert_allclose(dU1.cpu().numpy(), dUo, atol=3e-9, rtol=3e-4)
    np.testing.assert_allclose(dI1.cpu().numpy(), dIo, atol=3e-9, rtol=3e-4)


@pytest.mark.parametrize("B,d", [(40, 32), (300, 64), (257, 128)])
def test_inbatch_bf16x6_two_sweep_form_vs_oracle(B, d):
    """precision=2 without the stored G: user-mode AND item-mode sweeps of loss_x6.hip (the form used when G does not
    fit in HBM), at the f32 tolerances."""
    from recommendit_amd.two_tower import inbatch_loss_and_grads
    rng = np.random.RandomState(B + 5)
    U, I = fx.unit_rows(rng, B, d), fx.unit_rows(rng, B, d)
    loss, dU, dI = inbatch_loss_and_grads(t(U), t(I), precision=2, store_g=False)
    lo, dUo, dIo = O.in_batch_bpr_loss(U, I)
    assert abs(loss.item() - float(lo)) < 3e-6
    np.testing.assert_allclose(dU.cpu().numpy(), dUo, atol=3e-9, rtol=3e-4)
    np.testing.assert_allclose(dI.cpu().numpy(), dIo, atol=3e-9, rtol=3e-4)


@pytest.mark.parametrize("B,d", [(33, 32), (257, 64), (500, 128)])
def test_inbatch_stored_g_equals_recompute_form(B, d):
    """The stored-G form (user pass writes G, item pass = G^T.U) and the two-sweep form give the same dU/r bit for
    bit and the same dI up to f32 summation order; both agree with the oracle."""
    from recommendit_amd.two_tower import inbatch_loss_and_grads
    rng = np.random.RandomState(B + 1)
    U, I = fx.unit_rows(rng, B, d), fx.unit_rows(rng, B, d)
    l1, dU1, dI1 = inbatch_loss_and_grads(t(U), t(I), store_g=True)
    l2, dU2, dI2 = inbatch_loss_and_grads(t(U), t(I), store_g=False)
    assert torch.equal(l1, l2) and torch.equal(dU1, dU2)
    np.testing.assert_allclose(dI1.cpu().numpy(), dI2.cpu().numpy(), atol=1e-9, rtol=2e-5)
    _, _, dIo = O.in_batch_bpr_loss(U, I)
    np.testing.assert_allclose(dI1.cpu().numpy(), dIo, atol=3e-9, rtol=3e-4)


@pytest.mark.parametrize("Bl,G,off,d", [(100, 300, 100, 64), (64, 256, 192, 128), (37, 111, 0, 32),
                                         (4100, 4230, 77, 64)])   # last: 8-wave workgroups, ragged, split sweeps
@pytest.mark.parametrize("prec", [0, 2])
def test_inbatch_stored_g_rectangular_rank_form(Bl, G, off, d, prec):
    """Multi-GPU shape of the stored-G passes through the C ABI: local users [Bl] (positives = items off..off+Bl)
    against all G items; the item pass returns this rank's partial dI for ALL items (oracle: rectangular form)."""
    from recommendit_amd import _lib as L
    lib, dev, st = L.lib(), L.device(), L.stream_ptr()
    rng = np.random.RandomState(Bl)
    U, I = fx.unit_rows(rng, Bl, d), fx.unit_rows(rng, G, d)
    Ud, Id = t(U).to(dev).contiguous(), t(I).to(dev).contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    pos = torch.empty(Bl, **f32); r = torch.empty(Bl, **f32)
    dU = torch.empty(Bl, d, **f32); dI = torch.full((G, d), float("nan"), **f32)
    lp = torch.zeros(max(1024, lib.rihip_inbatch_workspace_doubles(Bl)), dtype=torch.float64, device=dev)
    ws = torch.empty(max(lib.rihip_inbatch_workspace_floats(Bl, G, d), lib.rihip_inbatch_workspace_floats(G, Bl, d)), **f32)
    gm = torch.full((lib.rihip_inbatch_gmat_floats(Bl, G),), float("nan"), **f32)   # unwritten blocks must not leak
    loss = torch.empty((), **f32)
    L.check(lib.rihip_rowdot(Ud.data_ptr(), Id.data_ptr(), Bl, off, d, pos.data_ptr(), st), "rowdot")
    L.check(lib.rihip_inbatch_user_pass(Ud.data_ptr(), Bl, off, Id.data_ptr(), G, 0, d, pos.data_ptr(), G,
                                        dU.data_ptr(), r.data_ptr(), lp.data_ptr(), ws.data_ptr(), gm.data_ptr(), prec, st), "up")
    L.check(lib.rihip_inbatch_item_pass(gm.data_ptr(), Ud.data_ptr(), Bl, off, G, 0, d, r.data_ptr(), G, dI.data_ptr(),
                                        ws.data_ptr(), prec, st), "ip")
    L.check(lib.rihip_sum_partials(lp.data_ptr(), lib.rihip_inbatch_loss_parts(Bl, G), 1.0 / (G * (G - 1.0)),
                                   loss.data_ptr(), st), "sum")
    lo, dUo, dIo = O.in_batch_bpr_loss(U, I, owner_offset=off, n_global=G)
    assert abs(loss.item() - float(lo)) < 3e-6
    np.testing.assert_allclose(dU.cpu().numpy(), dUo, atol=3e-9, rtol=3e-4)
    np.testing.assert_allclose(dI.cpu().numpy(), dIo, atol=3e-9, rtol=3e-4)


def test_inbatch_cfg4_per_rank_shape():
    """BASELINE.json configs[3] as ONE rank of eight sees it: 8 192 local users (positives = items off..off+8191 of
    the gathered batch) against all 65 536 all-gathered items, d = 128; the item pass returns this rank's partial dI
    for ALL items (what is reduce-scattered).  Oracle on slices: the first 64 local users against the rectangular
    closed form; 32 items inside and 32 outside the rank's own block against an fp64 G^T.U over the local users."""
    from recommendit_amd import _lib as L
    lib, dev, st = L.lib(), L.device(), L.stream_ptr()
    Bl, G, d, rank = 8192, 65536, 128, 3
    off = rank * Bl
    rng = np.random.RandomState(44)
    U, I = fx.unit_rows(rng, Bl, d), fx.unit_rows(rng, G, d)
    Ud, Id = t(U).to(dev).contiguous(), t(I).to(dev).contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    pos = torch.empty(Bl, **f32); r = torch.empty(Bl, **f32)
    dU = torch.empty(Bl, d, **f32); dI = torch.full((G, d), float("nan"), **f32)
    lp = torch.zeros(max(1024, lib.rihip_inbatch_workspace_doubles(Bl)), dtype=torch.float64, device=dev)
    ws = torch.empty(max(lib.rihip_inbatch_workspace_floats(Bl, G, d), lib.rihip_inbatch_workspace_floats(G, Bl, d)), **f32)
    gm = torch.empty((lib.rihip_inbatch_gmat_floats(Bl, G),), **f32)
    L.check(lib.rihip_rowdot(Ud.data_ptr(), Id.data_ptr(), Bl, off, d, pos.data_ptr(), st), "rowdot")
    L.check(lib.rihip_inbatch_user_pass(Ud.data_ptr(), Bl, off, Id.data_ptr(), G, 0, d, pos.data_ptr(), G,
                                        dU.data_ptr(), r.data_ptr(), lp.data_ptr(), ws.data_ptr(), gm.data_ptr(), 0, st), "up")
    L.check(lib.rihip_inbatch_item_pass(gm.data_ptr(), Ud.data_ptr(), Bl, off, G, 0, d, r.data_ptr(), G, dI.data_ptr(),
                                        ws.data_ptr(), 0, st), "ip")
    assert torch.isfinite(dI).all()
    _, dUo, _ = O.in_batch_bpr_loss(U[:64], I, owner_offset=off, n_global=G)
    np.testing.assert_allclose(dU[:64].cpu().numpy(), dUo, atol=1e-12, rtol=5e-4)
    U64 = U.astype(np.float64)
    posn = (U64 * I[off:off + Bl].astype(np.float64)).sum(1)[:, None]
    cols = np.concatenate([np.arange(off + 100, off + 132), np.arange(5, 37)])     # own block / another rank's items
    S = U64 @ I[cols].astype(np.float64).T
    Gm = 1.0 / (1.0 + np.exp(-(S - posn))) / (G * (G - 1.0))
    Gm[np.arange(100, 132), np.arange(32)] = 0.0                                  # the positives of users 100..131
    dI_off = Gm.T @ U64
    got = dI[t(cols).to(dev)].double().cpu().numpy()
    np.testing.assert_allclose(got[32:], dI_off[32:], atol=2e-11, rtol=5e-4)        # no diagonal term off-block
    res = got[:32] - dI_off[:32]                                                   # own block: -r_j u_j, r_j > 0
    coef = (res * U[100:132]).sum(1)
    assert (coef < 0).all()
    np.testing.assert_allclose(res, coef[:, None] * U[100:132], atol=2e-10, rtol=0)


def test_inbatch_unnormalised_inputs_keep_the_limits():
    """The element works in the log2 domain with sigma = 1/(1+2^-z2) and one log per 8 factors of (1+e^-z): rows of
    norm 2 (|z| up to 8) must still match the oracle, and saturated scores (|z| ~ 200) must give finite gradients
    (sigma -> 0 / 1) -- the loss may overflow there, by design."""
    from recommendit_amd.two_tower import inbatch_loss_and_grads
    rng = np.random.RandomState(5)
    B, d = 96, 32
    U, I = 2.0 * fx.unit_rows(rng, B, d), 2.0 * fx.unit_rows(rng, B, d)
    loss, dU, dI = inbatch_loss_and_grads(t(U), t(I))
    lo, dUo, dIo = O.in_batch_bpr_loss(U, I)
    assert abs(loss.item() - float(lo)) < 2e-5 * max(1.0, float(lo))
    np.testing.assert_allclose(dU.cpu().numpy(), dUo, atol=2e-8, rtol=1e-3)
    np.testing.assert_allclose(dI.cpu().numpy(), dIo, atol=2e-8, rtol=1e-3)
    _, dU2, dI2 = inbatch_loss_and_grads(t(10.0 * U), t(10.0 * I))
    assert torch.isfinite(dU2).all() and torch.isfinite(dI2).all()


def test_inbatch_full_size_properties():
    """cfg2 size (B=8192, d=64): size-independent properties instead of the O(B^2) oracle:
    sum_i dU_i . u_i + ... identities: d loss/d(scale) -- here: gradient of a loss that only depends on
    score differences is orthogonal to a uniform shift of all item vectors along any user direction, and
    sum over rows of G is zero => sum_j dI_j == sum_i G^T 1 ... we check  sum(dI) == sum_i (sum_j G_ij) u_i == 0."""
    from recommendit_amd.two_tower import inbatch_loss_and_grads
    rng = np.random.RandomState(0)
    B, d = 8192, 64
    U, I = fx.unit_rows(rng, B, d), fx.unit_rows(rng, B, d)
    loss, dU, dI = inbatch_loss_and_grads(t(U), t(I))
    assert 0.3 < loss.item() < 1.2
    # rows of G sum to zero  =>  dU_i . 1-shift: sum_j G_ij = 0  => sum_i dI_i = G^T U summed over j = U^T (G 1) = 0
    assert dI.sum(0).abs().max().item() < 1e-6
    # subset check against the oracle on the first 64 users (needs all items): rectangular oracle
    lo, dUo, _ = O.in_batch_bpr_loss(U[:64], I, owner_offset=0, n_global=B)
    np.testing.assert_allclose(dU[:64].cpu().numpy(), dUo, atol=1e-10, rtol=5e-4)


@pytest.mark.parametrize("prec", [0, 2])
def test_inbatch_headline_size_properties(prec):
    """BASELINE.json's headline shape (global batch 65 536, d = 128; 17 GB of stored G) through size-independent
    properties and oracle checks on slices: rows of G sum to zero (the item gradients sum to the null vector);
    <dI, I> == <dU, U> (both are sum_ij G_ij s_ij); the first 64 users against the rectangular oracle; the last 32
    items against an fp64 G^T.U over ALL users, whose residual must be the diagonal term -r_j u_j (r_j > 0)."""
    from recommendit_amd.two_tower import inbatch_loss_and_grads
    rng = np.random.RandomState(11)
    B, d = 65536, 128
    U, I = fx.unit_rows(rng, B, d), fx.unit_rows(rng, B, d)
    Ud, Id = t(U), t(I)
    loss, dU, dI = inbatch_loss_and_grads(Ud, Id, precision=prec, store_g=True)
    assert 0.6 < loss.item() < 0.8            # random unit vectors: softplus(z), z ~ N(0, 2/d)
    assert dI.sum(0).abs().max().item() < 2e-7
    # <dI, I> = sum_ij G_ij (u_i . i_j) = <dU, U>   (G includes the diagonal entries)
    a = (dI.double() * Id.to(dI.device).double()).sum().item()
    b = (dU.double() * Ud.to(dU.device).double()).sum().item()
    assert abs(a - b) <= 1e-6 * max(abs(a), abs(b), 1e-12) + 1e-12, (a, b)
    lo, dUo, _ = O.in_batch_bpr_loss(U[:64], I, owner_offset=0, n_global=B)
    np.testing.assert_allclose(dU[:64].cpu().numpy(), dUo, atol=1e-12, rtol=5e-4)
    # a column slice of dI against the dense oracle restricted to those items needs all users: do it for 32 items
    S = U.astype(np.float64) @ I[-32:].astype(np.float64).T                    # [B, 32]
    pos = (U.astype(np.float64) * I.astype(np.float64)).sum(1)[:, None]
    G = 1.0 / (1.0 + np.exp(-(S - pos))) / (B * (B - 1.0))
    G[np.arange(B - 32, B), np.arange(32)] = 0.0
    dI_off = G.T @ U.astype(np.float64)                                        # off-diagonal part of dI[-32:]
    # dI_j = dI_off_j - r_j u_j with r_j > 0: the residual must be a negative multiple of u_j
    res = dI[-32:].double().cpu().numpy() - dI_off
    coef = (res * U[-32:]).sum(1)
    assert (coef < 0).all()
    np.testing.assert_allclose(res, coef[:, None] * U[-32:], atol=2e-10, rtol=0)


def test_g4_train50_golden_with_stock_adam(golden_dir):
    """Drop-in check: stock torch.optim.Adam + clip_grad_norm_ + CosineAnnealingLR drive the HIP model
    exactly like the reference loop (train_embeddings.py:183-197)."""
    g = np.load(golden_dir / "g4_train50.npz")
    nu, ni, d, H, seed, B = (int(x) for x in g["cfg"])
    m, sd = _model(nu, ni, d, H, seed)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=2)
    for step in range(50):
        u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=4000 + step, boundary=False)
        U = m.user_tower(t(u)); P = m.item_tower(t(p), t(gp)); N = m.item_tower(t(n), t(gn))
        loss = m.bpr_loss(U, P, N)
        opt.zero_grad(); loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)
        opt.step()
        assert abs(loss.item() - g["losses"][step]) < 5e-5, step
        if step == 24:
            sched.step()
    for k, prm in m.named_parameters():
        np.testing.assert_allclose(prm.detach().cpu().numpy(), g[f"final_{k}"], atol=2e-4, rtol=0, err_msg=k)


def test_g5_g6_inference_and_reference_checkpoint(golden_dir, tmp_path, fwd_kernel):
    from recommendit_amd import TwoTowerModel
    g = np.load(golden_dir / "g5_inference.npz")
    nu, ni, d, H, seed = (int(x) for x in g["cfg"])
    m, sd = _model(nu, ni, d, H, seed)
    E = m.get_item_embeddings(list(range(1, 1001)), g["genres"])
    np.testing.assert_allclose(E, g["item_embs"], atol=2e-6)
    np.testing.assert_allclose(m.get_user_embedding(7), g["user7"], atol=2e-6)
    np.testing.assert_allclose(m.get_user_embedding(100), g["user100"], atol=2e-6)
    # a checkpoint written by the reference's save() loads and reproduces its outputs
    ref = TwoTowerModel.load(str(golden_dir / "g6_reference_checkpoint.pt"))
    exp = np.load(golden_dir / "g6_expected.npz")["U"]
    with torch.no_grad():
        got = ref.user_tower(torch.tensor([1, 2, 20])).numpy()
    np.testing.assert_allclose(got, exp, atol=2e-6)
    assert ref._item_id_to_idx == {1: 0, 2: 1, 3: 2}
    # own save/load round trip with a non-default hidden_dim (the reference cannot do this)
    m.precompute_item_embeddings([1, 2, 3], np.zeros((3, 18), np.float32))
    m.save(str(tmp_path / "tt.pt"))
    m2 = TwoTowerModel.load(str(tmp_path / "tt.pt"))
    assert (m2.n_users, m2.n_items, m2.embed_dim, m2.hidden_dim) == (nu, ni, d, H)
    np.testing.assert_allclose(m2.get_user_embedding(7), g["user7"], atol=2e-6)


@pytest.mark.parametrize("B,d", [(33, 32), (130, 64), (500, 128), (2048, 128)])
def test_inbatch_bf16x3_precision_mode(B, d):
    """Optional split-bf16 sweep (hi.hi + hi.lo + lo.hi on bf16 MFMA, f32 accumulate): same results as the exact
    path within the split's 2^-16 product error."""
    from recommendit_amd.two_tower import inbatch_loss_and_grads
    rng = np.random.RandomState(B + 1)
    U, I = fx.unit_rows(rng, B, d), fx.unit_rows(rng, B, d)
    l1, dU1, dI1 = inbatch_loss_and_grads(t(U), t(I), precision=1)
    l2, dU2, dI2 = inbatch_loss_and_grads(t(U), t(I), precision=1)
    assert torch.equal(dU1, dU2) and torch.equal(dI1, dI2) and torch.equal(l1, l2)   # still deterministic
    lo, dUo, dIo = O.in_batch_bpr_loss(U, I)
    assert abs(l1.item() - float(lo)) < 5e-6
    np.testing.assert_allclose(dU1.cpu().numpy(), dUo, atol=5e-9, rtol=1e-3)
    np.testing.assert_allclose(dI1.cpu().numpy(), dIo, atol=5e-9, rtol=1e-3)
    # and close to the exact-f32 kernel
    l0, dU0, dI0 = inbatch_loss_and_grads(t(U), t(I), precision=0)
    scale = dU0.abs().max().item()
    assert (dU1 - dU0).abs().max().item() < 2e-4 * scale


@pytest.mark.parametrize("B,d,H", [(200, 64, 128), (3000, 32, 64), (50000, 128, 128)])
def test_backward_in_two_halves_is_bitwise_the_single_call(B, d, H):
    """rihip_tower_backward_partial (gradient kernels, slabs left in the workspace) + rihip_tower_backward_reduce2 (the
    slabs of BOTH towers summed in one pair of launches) give bit-for-bit the gradients of two rihip_tower_backward
    calls -- same sums, same order; B = 50 000 takes the two-kernel backward with 256 slabs (two reduce levels)."""
    import ctypes as C
    from recommendit_amd import _lib as L
    lib, dev = L.lib(), L.device()
    g = torch.Generator(device=dev); g.manual_seed(B)
    n_rows = 5000
    f32 = dict(dtype=torch.float32, device=dev)

    def tower(item):
        K1 = d + (18 if item else 0)
        Bt = 2 * B if item else B
        return dict(
            item=item, B=Bt, table=torch.randn((n_rows, d), generator=g, **f32),
            ids=torch.randint(1, n_rows, (Bt,), device=dev, generator=g),
            genres=(torch.rand((Bt, 18), device=dev, generator=g) < 0.2).float() if item else None,
            W1=torch.randn((H, K1), generator=g, **f32) * 0.1, W2=torch.randn((d, H), generator=g, **f32) * 0.1,
            gout=torch.randn((Bt, d), generator=g, **f32), out=torch.randn((Bt, d), generator=g, **f32),
            den=torch.rand((Bt,), generator=g, **f32) + 0.5, hid=torch.relu(torch.randn((Bt, H), generator=g, **f32)),
            ws=torch.empty((lib.rihip_tower_backward_workspace_floats(Bt, d, H, 1 if item else 0),), **f32))

    def grads(t):
        K1 = d + (18 if t["item"] else 0)
        return [torch.empty((H, K1), **f32), torch.empty((H,), **f32), torch.empty((d, H), **f32), torch.empty((d,), **f32)]

    def common(t, dX):
        return (t["table"].data_ptr(), n_rows, t["ids"].data_ptr(), L.ptr(t["genres"]), t["B"], d, H, t["W1"].data_ptr(),
                t["W2"].data_ptr(), t["gout"].data_ptr(), t["out"].data_ptr(), t["den"].data_ptr(), t["hid"].data_ptr(),
                1.25, dX.data_ptr())

    tu, ti = tower(False), tower(True)
    st = L.stream_ptr()
    ref, dX_ref = [], []
    for t_ in (tu, ti):
        gr, dX = grads(t_), torch.empty((t_["B"], d), **f32)
        L.check(lib.rihip_tower_backward(*common(t_, dX), *(x.data_ptr() for x in gr), 0, t_["ws"].data_ptr(), st), "bwd")
        ref.append([x.clone() for x in gr]); dX_ref.append(dX.clone())
    got, dX_got, ns = [grads(tu), grads(ti)], [], []
    for t_ in (tu, ti):
        dX = torch.empty((t_["B"], d), **f32)
        n = C.c_int(0)
        L.check(lib.rihip_tower_backward_partial(*common(t_, dX), t_["ws"].data_ptr(), st, None, C.byref(n)), "partial")
        dX_got.append(dX); ns.append(n.value)
    assert ns[0] > 0 and ns[1] > 0
    L.check(lib.rihip_tower_backward_reduce2(d, H, tu["ws"].data_ptr(), tu["B"], 0, ns[0], *(x.data_ptr() for x in got[0]),
                                             ti["ws"].data_ptr(), ti["B"], 1, ns[1], *(x.data_ptr() for x in got[1]), 0, st),
            "reduce2")
    for a_, b_ in zip(dX_ref + ref[0] + ref[1], dX_got + got[0] + got[1]):
        assert torch.equal(a_, b_)


def test_clip_coef_step_folds_loss_sum_and_step_clock():
    """rihip_clip_coef_step = rihip_clip_coef + the Adam clock of rihip_adam_hyper_step (for the step that is running)
    + the loss sum of rihip_sum_partials, in one launch."""
    from recommendit_amd import _lib as L
    lib, dev = L.lib(), L.device()
    g = torch.Generator(device=dev); g.manual_seed(3)
    part = torch.rand((3000,), dtype=torch.float64, device=dev, generator=g)
    lpart = torch.rand((700,), dtype=torch.float64, device=dev, generator=g)
    f32 = dict(dtype=torch.float32, device=dev)
    coef, norm, hyper, loss = (torch.zeros((1,), **f32), torch.zeros((1,), **f32), torch.zeros((2,), **f32),
                               torch.zeros((1,), **f32))
    step = torch.full((1,), 7, dtype=torch.int64, device=dev)
    lr = torch.full((1,), 3e-3, **f32)
    st = L.stream_ptr()
    L.check(lib.rihip_clip_coef_step(part.data_ptr(), part.numel(), 1.0, coef.data_ptr(), norm.data_ptr(), step.data_ptr(),
                                     lr.data_ptr(), 0.9, 0.999, hyper.data_ptr(), lpart.data_ptr(), lpart.numel(), 0.25,
                                     loss.data_ptr(), st), "clip_coef_step")
    tn = float(np.sqrt(part.sum().item()))
    assert abs(norm.item() - tn) <= 1e-5 * tn and abs(coef.item() - min(1.0, 1.0 / (tn + 1e-6))) < 1e-6
    assert int(step.item()) == 8                                      # advanced for the next step
    b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))        # the ABI takes the betas as C floats
    assert abs(hyper[0].item() - 3e-3 / (1 - b1 ** 7)) < 1e-8 and abs(hyper[1].item() - np.sqrt(1 - b2 ** 7)) < 1e-7
    assert abs(loss.item() - 0.25 * lpart.sum().item()) < 1e-4
    # without loss partials the loss output is left alone
    loss.fill_(-1.0)
    L.check(lib.rihip_clip_coef_step(part.data_ptr(), part.numel(), 1.0, coef.data_ptr(), norm.data_ptr(), step.data_ptr(),
                                     lr.data_ptr(), 0.9, 0.999, hyper.data_ptr(), None, 0, 0.0, loss.data_ptr(), st), "clip")
    assert loss.item() == -1.0 and int(step.item()) == 9


@pytest.mark.parametrize("B,d,H", [(300, 64, 128), (5000, 32, 64), (70, 128, 128)])
def test_pair_launches_are_bitwise_the_single_tower_calls(B, d, H):
    """rihip_tower_forward_pair / rihip_tower_backward_partial_pair (user + item tower of a step in one launch; B = 5000
    also packs the four weight matrices in one launch) against the single-tower entry points: identical bits."""
    import ctypes as C
    from recommendit_amd import _lib as L
    lib, dev = L.lib(), L.device()
    g = torch.Generator(device=dev); g.manual_seed(B + d)
    f32 = dict(dtype=torch.float32, device=dev)
    n_rows, p_drop = 4000, 0.2
    step = torch.full((1,), 5, dtype=torch.int64, device=dev)
    err = torch.zeros((1,), dtype=torch.int32, device=dev)

    def tower(item):
        K1, Bt = d + (18 if item else 0), (2 * B if item else B)
        t_ = dict(item=item, B=Bt, table=torch.randn((n_rows, d), generator=g, **f32),
                  ids=torch.randint(1, n_rows, (Bt,), device=dev, generator=g),
                  genres=(torch.rand((Bt, 18), device=dev, generator=g) < 0.2).float() if item else None,
                  W1=torch.randn((H, K1), generator=g, **f32) * 0.1, b1=torch.randn((H,), generator=g, **f32) * 0.1,
                  W2=torch.randn((d, H), generator=g, **f32) * 0.1, b2=torch.randn((d,), generator=g, **f32) * 0.1,
                  gout=torch.randn((Bt, d), generator=g, **f32), seed=11 + item)
        for tag in ("a", "b"):   # a = single calls, b = pair calls
            t_[tag] = dict(out=torch.empty((Bt, d), **f32), hid=torch.empty((Bt, H), **f32), den=torch.empty((Bt,), **f32),
                           dX=torch.empty((Bt, d), **f32),
                           fws=torch.empty((lib.rihip_tower_forward_workspace_floats(d, H, int(item)),), **f32),
                           bws=torch.empty((lib.rihip_tower_backward_workspace_floats(Bt, d, H, int(item)),), **f32))
        return t_

    tu, ti = tower(False), tower(True)
    st = L.stream_ptr()
    nsl = {}
    for t_ in (tu, ti):
        o = t_["a"]
        L.check(lib.rihip_tower_forward(t_["table"].data_ptr(), n_rows, t_["ids"].data_ptr(), L.ptr(t_["genres"]), t_["B"], d, H,
                                        t_["W1"].data_ptr(), t_["b1"].data_ptr(), t_["W2"].data_ptr(), t_["b2"].data_ptr(), 1,
                                        p_drop, t_["seed"], 0, o["out"].data_ptr(), o["hid"].data_ptr(), o["den"].data_ptr(),
                                        err.data_ptr(), o["fws"].data_ptr(), step.data_ptr(), st), "fwd")
        n = C.c_int(0)
        L.check(lib.rihip_tower_backward_partial(t_["table"].data_ptr(), n_rows, t_["ids"].data_ptr(), L.ptr(t_["genres"]),
                                                 t_["B"], d, H, t_["W1"].data_ptr(), t_["W2"].data_ptr(), t_["gout"].data_ptr(),
                                                 o["out"].data_ptr(), o["den"].data_ptr(), o["hid"].data_ptr(), 1.25,
                                                 o["dX"].data_ptr(), o["bws"].data_ptr(), st, None, C.byref(n)), "bwd")
        nsl[t_["item"]] = n.value
    ios = []
    for t_ in (tu, ti):
        o, io = t_["b"], L.TowerIO()
        io.table, io.n_rows, io.ids, io.genres, io.B = t_["table"].data_ptr(), n_rows, t_["ids"].data_ptr(), L.ptr(t_["genres"]), t_["B"]
        io.W1, io.b1, io.W2, io.b2 = t_["W1"].data_ptr(), t_["b1"].data_ptr(), t_["W2"].data_ptr(), t_["b2"].data_ptr()
        io.seed, io.row0 = t_["seed"], 0
        io.out, io.hid, io.denom, io.fwd_workspace = o["out"].data_ptr(), o["hid"].data_ptr(), o["den"].data_ptr(), o["fws"].data_ptr()
        io.grad_out, io.dX, io.bwd_workspace = t_["gout"].data_ptr(), o["dX"].data_ptr(), o["bws"].data_ptr()
        ios.append(io)
    L.check(lib.rihip_tower_forward_pair(C.byref(ios[0]), C.byref(ios[1]), d, H, 1, p_drop, err.data_ptr(), step.data_ptr(), st),
            "fwd_pair")
    nu, ni = C.c_int(0), C.c_int(0)
    L.check(lib.rihip_tower_backward_partial_pair(C.byref(ios[0]), C.byref(ios[1]), d, H, 1.25, st, None, None, C.byref(nu),
                                                  C.byref(ni)), "bwd_pair")
    assert (nu.value, ni.value) == (nsl[False], nsl[True])
    for t_, ns in ((tu, nu.value), (ti, ni.value)):
        for key in ("out", "hid", "den", "dX"):
            assert torch.equal(t_["a"][key], t_["b"][key]), key
        K1 = d + (18 if t_["item"] else 0)
        P = H * K1 + H + d * H + d
        assert torch.equal(t_["a"]["bws"][: ns * P], t_["b"]["bws"][: ns * P])      # the weight-gradient slabs
    assert int(err.item()) == 0


@pytest.mark.parametrize("B", [300, 6000])
def test_reduce_and_scatter_in_one_launch_is_bitwise_the_two_calls(B):
    """rihip_backward_reduce2_scatter2 = rihip_tower_backward_reduce2 + rihip_embedding_scatter_add2 (B = 6000: 94 / 188
    slabs, i.e. the first reduction level rides in the scatter launch and the second follows)."""
    import ctypes as C
    from recommendit_amd import _lib as L
    lib, dev = L.lib(), L.device()
    d, H, n_rows = 64, 128, 3000
    g = torch.Generator(device=dev); g.manual_seed(B)
    f32 = dict(dtype=torch.float32, device=dev)
    st = L.stream_ptr()
    tw = []
    for item in (0, 1):
        Bt, K1 = B * (2 if item else 1), d + (18 if item else 0)
        t_ = dict(B=Bt, K1=K1, table=torch.randn((n_rows, d), generator=g, **f32),
                  ids=torch.randint(0, n_rows, (Bt,), device=dev, generator=g),
                  genres=(torch.rand((Bt, 18), device=dev, generator=g) < 0.2).float() if item else None,
                  W1=torch.randn((H, K1), generator=g, **f32) * 0.1, W2=torch.randn((d, H), generator=g, **f32) * 0.1,
                  gout=torch.randn((Bt, d), generator=g, **f32), out=torch.randn((Bt, d), generator=g, **f32),
                  den=torch.rand((Bt,), generator=g, **f32) + 0.5, hid=torch.relu(torch.randn((Bt, H), generator=g, **f32)),
                  dX=torch.empty((Bt, d), **f32),
                  ws=torch.empty((lib.rihip_tower_backward_workspace_floats(Bt, d, H, item),), **f32))
        n = C.c_int(0)
        L.check(lib.rihip_tower_backward_partial(t_["table"].data_ptr(), n_rows, t_["ids"].data_ptr(), L.ptr(t_["genres"]), Bt,
                                                 d, H, t_["W1"].data_ptr(), t_["W2"].data_ptr(), t_["gout"].data_ptr(),
                                                 t_["out"].data_ptr(), t_["den"].data_ptr(), t_["hid"].data_ptr(), 1.0,
                                                 t_["dX"].data_ptr(), t_["ws"].data_ptr(), st, None, C.byref(n)), "partial")
        t_["ns"] = n.value
        tw.append(t_)

    def outs():
        return [[torch.empty((H, t_["K1"]), **f32), torch.empty((H,), **f32), torch.empty((d, H), **f32),
                 torch.empty((d,), **f32)] for t_ in tw], [torch.full((n_rows, d), 0.25, **f32) for _ in tw]

    def red_args(gr):
        a = [d, H]
        for i, t_ in enumerate(tw):
            a += [t_["ws"].data_ptr(), t_["B"], i, t_["ns"]] + [x.data_ptr() for x in gr[i]]
        return a + [0]

    def scat_args(tabs):
        a = []
        for i, t_ in enumerate(tw):
            a += [tabs[i].data_ptr(), n_rows, t_["ids"].data_ptr(), t_["dX"].data_ptr(), t_["B"]]
        return a

    g1, t1 = outs()
    L.check(lib.rihip_tower_backward_reduce2(*red_args(g1), st), "reduce2")
    L.check(lib.rihip_embedding_scatter_add2(*scat_args(t1), d, st), "scatter2")
    g2, t2 = outs()
    L.check(lib.rihip_backward_reduce2_scatter2(*red_args(g2), *scat_args(t2), st), "fused")
    for a_, b_ in zip(g1[0] + g1[1] + t1, g2[0] + g2[1] + t2):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("item", [False, True])
@pytest.mark.parametrize("B", [50000, 65536 + 17, 31, 8192])
def test_one_kernel_backward_equals_the_two_kernel_form(B, item, monkeypatch):
    """tower3.hip (data + weight gradients of a tile in one kernel, no gy / dPre round trip) against tower2.hip's two
    kernels on the same inputs: dX and the four weight gradients agree to float-summation-order level; ragged B, a batch
    smaller than one tile, duplicate ids, dropout scale, rows whose normalisation clamped"""
    import ctypes as C
    from recommendit_amd import _lib as L
    lib, dev = L.lib(), L.device()
    d = H = 128
    g = torch.Generator(device=dev); g.manual_seed(B + int(item))
    n_rows = 4000
    f32 = dict(dtype=torch.float32, device=dev)
    K1 = d + (18 if item else 0)
    t_ = dict(table=torch.randn((n_rows, d), generator=g, **f32), ids=torch.randint(1, n_rows, (B,), device=dev, generator=g),
              genres=(torch.rand((B, 18), device=dev, generator=g) < 0.2).float() if item else None,
              W1=torch.randn((H, K1), generator=g, **f32) * 0.1, W2=torch.randn((d, H), generator=g, **f32) * 0.1,
              gout=torch.randn((B, d), generator=g, **f32), out=torch.randn((B, d), generator=g, **f32),
              den=torch.rand((B,), generator=g, **f32) + 0.5, hid=torch.relu(torch.randn((B, H), generator=g, **f32)))
    t_["den"][::97] = 1e-12                                       # the clamp branch of F.normalize
    res = {}
    for which in ("3", "4"):
        monkeypatch.setenv("RIHIP_TOWER_BWD", which)
        ws = torch.empty((lib.rihip_tower_backward_workspace_floats(B, d, H, int(item)),), **f32)
        gr = [torch.empty((H, K1), **f32), torch.empty((H,), **f32), torch.empty((d, H), **f32), torch.empty((d,), **f32)]
        dX = torch.full((B, d), 7.0, **f32)
        L.check(lib.rihip_tower_backward(t_["table"].data_ptr(), n_rows, t_["ids"].data_ptr(), L.ptr(t_["genres"]), B, d, H,
                                         t_["W1"].data_ptr(), t_["W2"].data_ptr(), t_["gout"].data_ptr(), t_["out"].data_ptr(),
                                         t_["den"].data_ptr(), t_["hid"].data_ptr(), 1.25, dX.data_ptr(),
                                         *(x.data_ptr() for x in gr), 0, ws.data_ptr(), L.stream_ptr()), "bwd")
        torch.cuda.synchronize()
        res[which] = [dX] + gr
    for name, a_, b_ in zip(("dX", "dW1", "db1", "dW2", "db2"), res["3"], res["4"]):
        scale = float(a_.abs().max()) + 1e-30
        assert float((a_ - b_).abs().max()) <= 2e-5 * scale, name
        assert torch.isfinite(b_).all(), name
