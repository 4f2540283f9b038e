"""GPU: edge cases the reference's tests and callers exercise -- empty and single-row batches, k > N, one query,
ids on the table boundary, repeated ids, status codes instead of aborts."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import two_tower_np as O

pytestmark = pytest.mark.gpu


def _model(nu=50, ni=60, d=32, H=64, seed=4):
    from recommendit_amd import TwoTowerModel
    sd = fx.make_state(nu, ni, d, H, seed)
    m = TwoTowerModel(nu, ni, embed_dim=d, hidden_dim=H, dropout=0.0)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return m, sd


def test_empty_and_single_row_batches():
    m, sd = _model()
    m.eval()
    with torch.no_grad():
        e = m.user_tower(torch.zeros((0,), dtype=torch.long))
        assert e.shape == (0, 32)
        one = m.item_tower(torch.tensor([60]), torch.ones((1, 18)))
    assert one.shape == (1, 32) and abs(float(one.norm()) - 1.0) < 1e-5
    assert m.get_item_embeddings([], np.zeros((0, 18), np.float32)).shape == (0, 32)


def test_repeated_and_boundary_ids_gradients():
    """All samples hit the same two rows (first and last of the table): the dense scatter-add and the row-sparse
    segment reduce must both sum 257 contributions per row."""
    from recommendit_amd.trainer import HipBPRTrainer
    nu, ni, d, H, B = 50, 60, 32, 64, 257
    m, sd = _model(nu, ni, d, H)
    m.train()
    u = np.where(np.arange(B) % 2 == 0, 1, nu).astype(np.int64)
    p = np.where(np.arange(B) % 2 == 0, ni, 1).astype(np.int64)
    g = np.zeros((B, 18), np.float32)
    U = m.user_tower(torch.from_numpy(u).cuda()); P = m.item_tower(torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda())
    loss = m.in_batch_bpr_loss(U, P)
    loss.backward()
    gu = m.user_tower.embedding.weight.grad.cpu().numpy()
    assert np.abs(gu[2:nu]).max() == 0 and np.abs(gu[0]).max() == 0 and np.abs(gu[1]).max() > 0 and np.abs(gu[nu]).max() > 0
    pu = O.TowerParams(sd["user_tower.embedding.weight"], sd["user_tower.mlp.0.weight"], sd["user_tower.mlp.0.bias"],
                       sd["user_tower.mlp.3.weight"], sd["user_tower.mlp.3.bias"])
    pi = O.TowerParams(sd["item_tower.embedding.weight"], sd["item_tower.mlp.0.weight"], sd["item_tower.mlp.0.bias"],
                       sd["item_tower.mlp.3.weight"], sd["item_tower.mlp.3.bias"])
    Uo, cu = O.tower_forward(pu, u); Po, cp = O.tower_forward(pi, p, g)
    lo, dU, dI = O.in_batch_bpr_loss(Uo, Po)
    ref = O.embedding_scatter_add(nu + 1, u, O.tower_backward(pu, cu, dU)[0])
    np.testing.assert_allclose(gu, ref, atol=1e-8, rtol=2e-3)
    # the sparse path groups the same ids
    m2, _ = _model(nu, ni, d, H)
    m2.train()
    tr = HipBPRTrainer(m2, B, lr=1e-3, loss_mode="inbatch", table_opt="sparse")
    before = m2.user_tower.embedding.weight.detach().clone()
    tr.step(torch.from_numpy(u).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda())
    moved = (m2.user_tower.embedding.weight.detach() - before).abs().sum(1).cpu().numpy()
    assert moved[1] > 0 and moved[nu] > 0 and moved[2:nu].max() == 0 and moved[0] == 0


def test_search_k_larger_than_n_and_single_query():
    from recommendit_amd import FAISSIndex
    rng = np.random.RandomState(0)
    X = fx.unit_rows(rng, 37, 32)
    idx = FAISSIndex(embed_dim=32, exact=True)
    idx.build_ivf_index(X, list(range(100, 137)))
    d, ids = idx.search(X[3] * 7.0, k=500)
    assert len(ids) == 37 and ids[0] == 103 and abs(d[0] - 1.0) < 1e-6 and (np.diff(d) <= 0).all()
    s, r = idx.batch_search(X[:1], k=5)
    assert s.shape == (1, 5) and r[0, 0] == 100


def test_status_codes_not_aborts():
    from recommendit_amd import _lib
    l = _lib.lib()
    h = ctypes.c_void_p()
    assert l.rihip_ip_index_create(48, ctypes.byref(h)) == 0 and l.rihip_ip_index_destroy(h) == 0   # any width up to 128
    assert l.rihip_ip_index_create(200, ctypes.byref(h)) != 0 and b"unsupported" in l.rihip_last_error()
    assert l.rihip_gbdt_load_text(b"/nonexistent/model.txt", ctypes.byref(h)) != 0 and b"cannot open" in l.rihip_last_error()
    x = torch.zeros(8, device="cuda")
    assert l.rihip_adam_dense(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0,
                              None, None, None) != 0  # step must be >= 1 when no device clock is given
    with pytest.raises(RuntimeError, match="no HIP kernel instantiation|unsupported"):
        from recommendit_amd import TwoTowerModel
        TwoTowerModel(5, 5, embed_dim=40, hidden_dim=64).user_tower(torch.tensor([1]))   # not a multiple of 16


def test_multi_tensor_launches_equal_single_tensor_calls():
    """rihip_sumsq_multi / rihip_adam_dense_multi are the same arithmetic as three single-tensor calls, bit for bit,
    and the zero_grad mask clears exactly the flagged gradients."""
    from recommendit_amd import _lib as L
    lib, dev, st = L.lib(), L.device(), L.stream_ptr()
    g = torch.Generator(device="cpu").manual_seed(3)
    sizes = [4099, 640, 12800]
    P = [torch.randn(n, generator=g).to(dev) for n in sizes]
    G = [torch.randn(n, generator=g).to(dev) * 1e-2 for n in sizes]
    M = [torch.rand(n, generator=g).to(dev) * 1e-3 for n in sizes]
    V = [torch.rand(n, generator=g).to(dev) * 1e-5 for n in sizes]
    npart = lib.rihip_sumsq_nparts()
    part1 = torch.zeros(3 * npart, dtype=torch.float64, device=dev)
    part2 = torch.zeros_like(part1)
    for t in range(3):
        L.check(lib.rihip_sumsq(G[t].data_ptr(), sizes[t], part1.data_ptr() + 8 * t * npart, st), "sumsq")
    PA, NA = ctypes.c_void_p * 3, ctypes.c_int64 * 3
    L.check(lib.rihip_sumsq_multi(3, PA(*[x.data_ptr() for x in G]), NA(*sizes), part2.data_ptr(), st), "sumsq_multi")
    assert torch.equal(part1, part2)
    coef = torch.full((1,), 0.7, dtype=torch.float32, device=dev)
    P1, M1, V1 = [x.clone() for x in P], [x.clone() for x in M], [x.clone() for x in V]
    for t in range(3):
        L.check(lib.rihip_adam_dense(P1[t].data_ptr(), G[t].data_ptr(), M1[t].data_ptr(), V1[t].data_ptr(), sizes[t],
                                     1e-3, 0.9, 0.999, 1e-8, 1e-5, 7, coef.data_ptr(), None, st), "adam_dense")
    G2 = [x.clone() for x in G]
    L.check(lib.rihip_adam_dense_multi(3, PA(*[x.data_ptr() for x in P]), PA(*[x.data_ptr() for x in G2]),
                                       PA(*[x.data_ptr() for x in M]), PA(*[x.data_ptr() for x in V]), NA(*sizes), 0b101,
                                       1e-3, 0.9, 0.999, 1e-8, 1e-5, 7, coef.data_ptr(), None, st), "adam_dense_multi")
    for t in range(3):
        assert torch.equal(P[t], P1[t]) and torch.equal(M[t], M1[t]) and torch.equal(V[t], V1[t])
    assert float(G2[0].abs().max()) == 0.0 and float(G2[2].abs().max()) == 0.0 and torch.equal(G2[1], G[1])


def test_rows_group_key_bits_hint_is_equivalent():
    """Sorting only the significant id bits (n_rows hint) groups exactly like the full 64-bit sort."""
    from recommendit_amd import _lib as L
    lib, dev, st = L.lib(), L.device(), L.stream_ptr()
    B, d, n_rows = 5000, 32, 1000
    g = torch.Generator(device="cpu").manual_seed(9)
    ids = torch.randint(0, n_rows, (B,), generator=g).to(dev)
    dX = torch.randn(B, d, generator=g).to(dev)
    outs = []
    for hint in (0, n_rows, 1 << 40):
        ws = torch.empty(lib.rihip_rows_workspace_bytes(B, d), dtype=torch.uint8, device=dev)
        uniq = torch.empty(B, dtype=torch.int64, device=dev)
        Gc = torch.zeros(B, d, dtype=torch.float32, device=dev)
        part = torch.zeros(lib.rihip_rows_nparts(), dtype=torch.float64, device=dev)
        L.check(lib.rihip_rows_group(ids.data_ptr(), B, d, hint, uniq.data_ptr(), ws.data_ptr(), ws.numel(), st), "group")
        L.check(lib.rihip_rows_reduce(dX.data_ptr(), B, d, uniq.data_ptr(), ws.data_ptr(), Gc.data_ptr(), part.data_ptr(),
                                      st), "reduce")
        nu = int(torch.unique(ids).numel())
        outs.append((uniq[:nu].clone(), Gc[:nu].clone(), part.clone()))
    for u, gc, pt in outs[1:]:
        assert torch.equal(u, outs[0][0]) and torch.equal(gc, outs[0][1]) and torch.equal(pt, outs[0][2])
    ref = torch.zeros(n_rows, d, device=dev).index_add_(0, ids, dX)
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), ref[outs[0][0]].cpu().numpy() * (outs[0][0] != 0).float()[:, None].cpu().numpy(),
                               atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("B,n_rows,d", [(1, 5, 32), (256, 40, 64), (8192, 3953, 64), (40000, 300, 128), (777, 97, 20),
                                        (5000, 700_000, 32)])
def test_dense_scatter_is_sequential_sum_bitwise(B, n_rows, d):
    """rihip_embedding_scatter_add: every row receives its samples in batch order on top of its current contents --
    the float32 result equals np.add.at (sequential) bit for bit, twice in a row (no floating-point atomics), with ids
    outside [1, n_rows) skipped.  B = 40000 crosses the 16384-position pass; 700k rows need > 2 owners of 2^18 rows; the
    skewed cases take the LDS-staged path for hot rows."""
    from recommendit_amd import _lib as L
    lib, dev = L.lib(), L.device()
    rng = np.random.default_rng(B + d)
    ids = rng.integers(0, n_rows, size=B).astype(np.int64)     # includes padding id 0 and heavy repeats
    if B == 8192:                                                # skewed batch: hot rows with hundreds of samples
        ids = np.minimum(rng.zipf(1.05, size=B), n_rows - 1).astype(np.int64)
    if B == 40000:                                               # one row takes a third of the batch
        ids[rng.random(B) < 0.33] = 17
    if B > 10:
        ids[3] = n_rows + 7; ids[5] = -2                        # out of range: skipped
        ids[7] = n_rows - 1; ids[8] = 1                         # table boundary rows
    dX = (rng.standard_normal((B, d)) * 10.0 ** rng.integers(-6, 1, size=(B, 1))).astype(np.float32)
    g0 = rng.standard_normal((n_rows, d)).astype(np.float32) if n_rows < 10000 else np.zeros((n_rows, d), np.float32)
    ref = g0.copy()
    ok = (ids > 0) & (ids < n_rows)
    np.add.at(ref, ids[ok], dX[ok])
    tid, tdx = torch.from_numpy(ids).to(dev), torch.from_numpy(dX).to(dev)
    outs = []
    for _ in range(2):
        g = torch.from_numpy(g0).to(dev)
        L.check(lib.rihip_embedding_scatter_add(g.data_ptr(), n_rows, tid.data_ptr(), tdx.data_ptr(), B, d,
                                                L.stream_ptr()), "scatter")
        outs.append(g.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0].view(np.uint32), ref.view(np.uint32))


def test_dense_scatter_two_tables_one_call():
    from recommendit_amd import _lib as L
    lib, dev = L.lib(), L.device()
    rng = np.random.default_rng(12)
    d = 64
    spec = [(300, 50), (20000, 900)]
    dev_in, refs = [], []
    for B, n in spec:
        ids = rng.integers(0, n, size=B).astype(np.int64)
        dX = rng.standard_normal((B, d)).astype(np.float32)
        ref = np.full((n, d), 0.5, np.float32)                  # += onto existing contents
        np.add.at(ref, ids[ids > 0], dX[ids > 0])
        dev_in.append((torch.from_numpy(ids).to(dev), torch.from_numpy(dX).to(dev), torch.full((n, d), 0.5, device=dev)))
        refs.append(ref)
    (ia, xa, ga), (ib, xb, gb) = dev_in
    L.check(lib.rihip_embedding_scatter_add2(ga.data_ptr(), spec[0][1], ia.data_ptr(), xa.data_ptr(), spec[0][0],
                                             gb.data_ptr(), spec[1][1], ib.data_ptr(), xb.data_ptr(), spec[1][0],
                                             d, L.stream_ptr()), "scatter2")
    assert np.array_equal(ga.cpu().numpy(), refs[0]) and np.array_equal(gb.cpu().numpy(), refs[1])
