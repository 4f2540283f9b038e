"""GPU parity: the fused training step (no autograd) against the golden 50-step run of the reference
loop (G4) and against the NumPy oracle for the sparse-row / in-batch variants."""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import two_tower_np as O

pytestmark = pytest.mark.gpu


def t(x):
    return torch.from_numpy(np.asarray(x)).cuda()


def _model(nu, ni, d, H, seed, dropout=0.0):
    from recommendit_amd import TwoTowerModel
    sd = fx.make_state(nu, ni, d, H, seed)
    m = TwoTowerModel(nu, ni, embed_dim=d, hidden_dim=H, dropout=dropout)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return m, sd


def _params(sd, tower):
    return O.TowerParams(sd[f"{tower}.embedding.weight"], sd[f"{tower}.mlp.0.weight"], sd[f"{tower}.mlp.0.bias"],
                         sd[f"{tower}.mlp.3.weight"], sd[f"{tower}.mlp.3.bias"])


@pytest.mark.parametrize("persistent", [False, True])
@pytest.mark.parametrize("use_graph", [False, True])
def test_g4_train50_golden_fused_dense(golden_dir, use_graph, persistent):
    """use_graph=True: the step is captured into a hipGraph on the second call and replayed 48 times; the device-side
    Adam clock and the lr scalar must keep the run on the reference's trajectory (incl. the mid-run lr change).
    persistent=True: the whole step is ONE launch (csrc/step_persistent.hip, three grid barriers); False: the seven
    dependent launches."""
    from recommendit_amd.trainer import HipBPRTrainer, cosine_lr
    g = np.load(golden_dir / "g4_train50.npz")
    nu, ni, d, H, seed, B = (int(x) for x in g["cfg"])
    m, sd = _model(nu, ni, d, H, seed)
    m.train()
    tr = HipBPRTrainer(m, B, lr=1e-2, weight_decay=1e-5, loss_mode="sampled", table_opt="dense", use_graph=use_graph,
                       persistent=persistent)
    assert tr.persistent == persistent
    epoch = 0
    for step in range(50):
        u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=4000 + step, boundary=False)
        loss = tr.step(t(u), t(np.concatenate([p, n])), t(np.concatenate([gp, gn])), lr=cosine_lr(1e-2, epoch, 2))
        assert abs(loss.item() - g["losses"][step]) < 5e-5, step
        if step == 24:
            epoch += 1
    assert (tr._graph is not None) == use_graph
    tr.check_errors()
    for k, prm in m.named_parameters():   # module parameters are views of the trainer's buffers
        np.testing.assert_allclose(prm.detach().cpu().numpy(), g[f"final_{k}"], atol=2e-4, rtol=0, err_msg=k)


@pytest.mark.parametrize("cfg", [(6040, 3952, 64, 128, 256), (300, 500, 64, 128, 1024), (90, 70, 48, 96, 33),
                                 (200, 50, 128, 128, 2048), (50, 40, 32, 64, 1)])
def test_persistent_step_equals_the_multi_launch_step(cfg):
    """one persistent launch vs the seven dependent launches on the same batches, dropout ON (same counter-based masks):
    losses, clip coefficient and every parameter after 6 steps agree to float-summation-order level; duplicates in
    the batch (few items, many samples) exercise the in-order scatter"""
    from recommendit_amd.trainer import HipBPRTrainer
    nu, ni, d, H, B = cfg
    outs = []
    for persistent in (False, True):
        m, sd = _model(nu, ni, d, H, seed=3, dropout=0.1)
        m.train()
        tr = HipBPRTrainer(m, B, lr=5e-3, weight_decay=1e-5, loss_mode="sampled", table_opt="dense", seed=5,
                           persistent=persistent)
        losses, norms, first = [], [], None
        for step in range(6):
            u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=step * 7 + 1, boundary=False)
            losses.append(tr.step(t(u), t(np.concatenate([p, n])), t(np.concatenate([gp, gn]))).item())
            norms.append(tr.gnorm.item())
            if step == 0:
                first = {k: v.detach().cpu().numpy().copy() for k, v in m.named_parameters()}
        tr.check_errors()
        assert int(tr.step_dev.item()) == 7
        assert float(tr.uopt.grad.abs().max()) == 0.0 and float(tr.iopt.grad.abs().max()) == 0.0   # left zeroed
        outs.append((losses, norms, first, {k: v.detach().cpu().numpy().copy() for k, v in m.named_parameters()}))
    lr = 5e-3
    # step 1 starts from identical state: float-summation-order agreement of loss, gradient norm and every parameter
    # (an element whose gradient sits at Adam's eps = 1e-8 may move by a visible fraction of one lr-sized step: <= 0.1 %
    # of the elements, bounded by lr)
    assert abs(outs[1][0][0] - outs[0][0][0]) < 1e-6 and abs(outs[1][1][0] / outs[0][1][0] - 1) < 2e-5
    for k in outs[0][2]:
        a, b = outs[0][2][k], outs[1][2][k]
        np.testing.assert_allclose(b, a, atol=1.01 * lr, rtol=0, err_msg=k)
        assert np.sum(np.abs(a - b) > 3e-5) <= max(1, int(1e-3 * a.size)), k
    # later steps compound that sensitivity: the two trajectories stay together in loss and norm, and all but a few
    # elements of the parameters agree
    np.testing.assert_allclose(outs[1][0], outs[0][0], atol=2e-5, rtol=0)
    np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=2e-3)
    for k in outs[0][3]:
        a, b = outs[0][3][k], outs[1][3][k]
        np.testing.assert_allclose(b, a, atol=6.1 * lr, rtol=0, err_msg=k)
        assert float(np.mean(np.abs(a - b))) < 1e-4, k


def test_graph_replay_equals_eager_sparse_inbatch():
    """row-sparse optimiser (rocPRIM sort/scan inside the captured region) + in-batch sweep: graph == eager, bitwise."""
    from recommendit_amd.trainer import HipBPRTrainer
    nu, ni, d, H, B = 300, 150, 64, 128, 96
    outs = []
    for use_graph in (False, True):
        m, sd = _model(nu, ni, d, H, seed=3, dropout=0.2)
        m.train()
        tr = HipBPRTrainer(m, B, lr=5e-3, loss_mode="inbatch", table_opt="sparse", seed=7, use_graph=use_graph)
        losses = []
        for step in range(6):
            u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=step * 13, boundary=False)
            losses.append(tr.step(t(u), t(p), t(gp)).item())
        outs.append((losses, {k: v.detach().clone() for k, v in m.named_parameters()}))
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def _oracle_step(sd, mom, u, p, gp, n, gn, step, lr, mode, sparse):
    nu1, ni1 = sd["user_tower.embedding.weight"].shape[0], sd["item_tower.embedding.weight"].shape[0]
    pu, pi = _params(sd, "user_tower"), _params(sd, "item_tower")
    U, cu = O.tower_forward(pu, u)
    if mode == "sampled":
        P, cp = O.tower_forward(pi, p, gp); N, cn = O.tower_forward(pi, n, gn)
        loss, dU, dP, dN = O.bpr_loss(U, P, N)
        bu = O.tower_backward(pu, cu, dU); bp = O.tower_backward(pi, cp, dP); bn = O.tower_backward(pi, cn, dN)
        gi_dense = O.embedding_scatter_add(ni1, p, bp[0]) + O.embedding_scatter_add(ni1, n, bn[0])
        imlp = [bp[j] + bn[j] for j in range(1, 5)]
    else:
        P, cp = O.tower_forward(pi, p, gp)
        loss, dU, dP = O.in_batch_bpr_loss(U, P)
        bu = O.tower_backward(pu, cu, dU); bp = O.tower_backward(pi, cp, dP)
        gi_dense = O.embedding_scatter_add(ni1, p, bp[0])
        imlp = [bp[j] for j in range(1, 5)]
    gu_dense = O.embedding_scatter_add(nu1, u, bu[0])
    grads = {"user_tower.embedding.weight": gu_dense, "item_tower.embedding.weight": gi_dense,
             "user_tower.mlp.0.weight": bu[1], "user_tower.mlp.0.bias": bu[2], "user_tower.mlp.3.weight": bu[3],
             "user_tower.mlp.3.bias": bu[4], "item_tower.mlp.0.weight": imlp[0], "item_tower.mlp.0.bias": imlp[1],
             "item_tower.mlp.3.weight": imlp[2], "item_tower.mlp.3.bias": imlp[3]}
    c, _ = O.clip_coef([grads[k] for k in fx.PARAM_ORDER])
    for k in fx.PARAM_ORDER:
        if sparse and k.endswith("embedding.weight"):
            ids = np.unique(u) if k.startswith("user") else np.unique(np.concatenate([p, n]) if mode == "sampled" else p)
            ids = ids[ids != 0]
            O.adam_rows_sparse(sd[k], mom[0][k], mom[1][k], ids, grads[k][ids], step, lr, wd=1e-5, clip=c)
        else:
            O.adam_step(sd[k], grads[k], mom[0][k], mom[1][k], step, lr, wd=1e-5, clip=c)
    return float(loss)


@pytest.mark.parametrize("mode,opt,cfg,prec", [("sampled", "sparse", (100, 200, 32, 64, 48), 0),
                                               ("inbatch", "dense", (100, 200, 64, 128, 40), 0),
                                               ("inbatch", "sparse", (300, 150, 128, 128, 100), 0),
                                               ("inbatch", "sparse", (300, 150, 128, 128, 100), 2),   # bf16x6 passes
                                               ("inbatch", "dense", (100, 200, 64, 128, 40), 2),
                                               ("sampled", "sparse", (60, 30, 64, 128, 256), 0)])  # heavy duplicates
def test_fused_step_variants_vs_oracle(mode, opt, cfg, prec):
    from recommendit_amd.trainer import HipBPRTrainer
    nu, ni, d, H, B = cfg
    m, sd = _model(nu, ni, d, H, seed=3)
    m.train()
    tr = HipBPRTrainer(m, B, lr=5e-3, weight_decay=1e-5, loss_mode=mode, table_opt=opt, inbatch_precision=prec)
    mom = ({k: np.zeros_like(v) for k, v in sd.items()}, {k: np.zeros_like(v) for k, v in sd.items()})
    for step in range(1, 6):
        u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=step * 11, boundary=False)
        if mode == "sampled":
            loss = tr.step(t(u), t(np.concatenate([p, n])), t(np.concatenate([gp, gn])))
        else:
            loss = tr.step(t(u), t(p), t(gp))
        lo = _oracle_step(sd, mom, u, p, gp, n, gn, step, 5e-3, mode, opt == "sparse")
        assert abs(loss.item() - lo) < 1e-5, (step, loss.item(), lo)
    # in-batch gradients are O(1/B^2): elements near Adam's eps=1e-8 turn ulp-level gradient differences into
    # visible update differences (a few % of one lr-sized step), hence the looser bound for that mode
    atol = 3e-5 if mode == "sampled" else 2.5e-4
    for k, prm in m.named_parameters():
        got = prm.detach().cpu().numpy()
        np.testing.assert_allclose(got, sd[k], atol=atol, rtol=0, err_msg=k)
        # (one element of a 128-long bias may sit in Adam's eps region; more than that is a real difference)
        assert np.sum(np.abs(got - sd[k]) > 3e-5) <= max(1, int(1e-3 * got.size)), k


def test_device_sampler_invariants_and_short_training():
    """UserItemDataset invariants (reference train_embeddings.py:43-63) + the trainer learns on ML-1M-shaped data."""
    from recommendit_amd.synthetic import ml1m_like
    from recommendit_amd.train_embeddings import EmbeddingTrainer, UserItemDataset, build_item_genre_dict, retrieval_ndcg
    ratings, movies, gm = ml1m_like(n_users=600, n_item_ids=500, n_catalog=480, n_ratings=60000, seed=1)
    gd = build_item_genre_dict(movies)
    ds = UserItemDataset(ratings, gd, sorted(movies["item_id"].tolist()))
    assert len(ds) == int((ratings["rating"] >= 4).sum())
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    rated = set((ratings["user_id"] * 10000 + ratings["item_id"]).tolist())
    catalog = set(movies["item_id"].tolist())
    seen = 0
    for u, items, genres in ds.epoch_batches(1024, gen):
        B = u.numel()
        assert items.numel() == 2 * B and genres.shape == (2 * B, 18)
        un, neg = u.cpu().numpy(), items[B:].cpu().numpy()
        assert all(int(a) * 10000 + int(b) not in rated for a, b in zip(un, neg))
        assert set(neg.tolist()) <= catalog
        np.testing.assert_array_equal(genres.cpu().numpy(), gm[items.cpu().numpy()])
        seen += B
    assert seen == len(ds) // 1024 * 1024
    # host path stays reference-compatible
    s = ds[0]
    assert len(s) == 5 and s[2].shape == (18,) and int(s[0]) * 10000 + int(s[3]) not in rated

    tr = EmbeddingTrainer(model_output_path="/tmp/rihip_test/tt.pt", embed_dim=32, epochs=3, batch_size=512,
                          learning_rate=5e-3, dropout=0.1, seed=0)
    model = tr.train(ratings, movies)
    assert tr.history[-1]["loss"] < tr.history[0]["loss"] < 0.70
    res = retrieval_ndcg(model, ratings, movies)
    assert 0.0 <= res["ndcg@10"] <= 1.0 and res["n_users"] == 200
