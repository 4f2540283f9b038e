"""CPU: the NumPy oracle against the golden vectors captured from the reference
(oracle/make_golden.py) and the reference's own known-answer tests."""
import numpy as np
import pytest

from oracle import fixtures as fx
from oracle import metrics_np as M
from oracle import two_tower_np as O


def _params(sd, tower):
    return O.TowerParams(sd[f"{tower}.embedding.weight"], sd[f"{tower}.mlp.0.weight"], sd[f"{tower}.mlp.0.bias"],
                         sd[f"{tower}.mlp.3.weight"], sd[f"{tower}.mlp.3.bias"])


@pytest.mark.parametrize("tag", ["small", "ml1m", "d128"])
def test_g1_tower_forward(golden_dir, tag):
    g = np.load(golden_dir / "g1_tower_forward.npz")
    nu, ni, d, H, seed = (int(x) for x in g[f"{tag}_cfg"])
    sd = fx.make_state(nu, ni, d, H, seed)
    assert bytes(g[f"{tag}_sha"]).hex() == fx.state_checksum(sd)
    for B in (1, 16, 256):
        u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=100 + B)
        U, _ = O.tower_forward(_params(sd, "user_tower"), u)
        P, _ = O.tower_forward(_params(sd, "item_tower"), p, gp)
        np.testing.assert_allclose(U, g[f"{tag}_B{B}_U"], atol=2e-6, rtol=0)
        np.testing.assert_allclose(P, g[f"{tag}_B{B}_P"], atol=2e-6, rtol=0)


@pytest.mark.parametrize("tag", ["small", "mid"])
def test_g2_bpr_grads(golden_dir, tag):
    g = np.load(golden_dir / "g2_bpr_grads.npz")
    nu, ni, d, H, seed, B = (int(x) for x in g[f"{tag}_cfg"])
    sd = fx.make_state(nu, ni, d, H, seed)
    u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=200 + B)
    pu, pi = _params(sd, "user_tower"), _params(sd, "item_tower")
    U, cu = O.tower_forward(pu, u)
    P, cp = O.tower_forward(pi, p, gp)
    N, cn = O.tower_forward(pi, n, gn)
    loss, dU, dP, dN = O.bpr_loss(U, P, N)
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-6
    np.testing.assert_allclose(dU, g[f"{tag}_dU"], atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(dP, g[f"{tag}_dP"], atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(dN, g[f"{tag}_dN"], atol=1e-7, rtol=1e-4)
    dxu, dW1u, db1u, dW2u, db2u = O.tower_backward(pu, cu, dU)
    dxp, dW1p, db1p, dW2p, db2p = O.tower_backward(pi, cp, dP)
    dxn, dW1n, db1n, dW2n, db2n = O.tower_backward(pi, cn, dN)
    tol = dict(atol=2e-7, rtol=2e-4)
    np.testing.assert_allclose(O.embedding_scatter_add(nu + 1, u, dxu), g[f"{tag}_grad_user_tower.embedding.weight"], **tol)
    ge = O.embedding_scatter_add(ni + 1, p, dxp) + O.embedding_scatter_add(ni + 1, n, dxn)
    np.testing.assert_allclose(ge, g[f"{tag}_grad_item_tower.embedding.weight"], **tol)
    np.testing.assert_allclose(dW1u, g[f"{tag}_grad_user_tower.mlp.0.weight"], **tol)
    np.testing.assert_allclose(db1u, g[f"{tag}_grad_user_tower.mlp.0.bias"], **tol)
    np.testing.assert_allclose(dW2u, g[f"{tag}_grad_user_tower.mlp.3.weight"], **tol)
    np.testing.assert_allclose(db2u, g[f"{tag}_grad_user_tower.mlp.3.bias"], **tol)
    np.testing.assert_allclose(dW1p + dW1n, g[f"{tag}_grad_item_tower.mlp.0.weight"], **tol)
    np.testing.assert_allclose(db1p + db1n, g[f"{tag}_grad_item_tower.mlp.0.bias"], **tol)
    np.testing.assert_allclose(dW2p + dW2n, g[f"{tag}_grad_item_tower.mlp.3.weight"], **tol)
    np.testing.assert_allclose(db2p + db2n, g[f"{tag}_grad_item_tower.mlp.3.bias"], **tol)


@pytest.mark.parametrize("B", [2, 16, 256, 96])
def test_g3_inbatch(golden_dir, B):
    g = np.load(golden_dir / "g3_inbatch.npz")
    U, I = g[f"B{B}_U"], g[f"B{B}_I"]
    loss, dU, dI = O.in_batch_bpr_loss(U, I)
    assert abs(float(loss) - float(g[f"B{B}_loss"])) < 2e-6
    np.testing.assert_allclose(dU, g[f"B{B}_dU"], atol=2e-8, rtol=2e-4)
    np.testing.assert_allclose(dI, g[f"B{B}_dI"], atol=2e-8, rtol=2e-4)
    if B <= 16:
        assert abs(float(O.in_batch_bpr_loss_loop(U, I)) - float(g[f"B{B}_loss"])) < 2e-6


def test_inbatch_rectangular_sums_to_square():
    rng = np.random.RandomState(5)
    U, I = fx.unit_rows(rng, 24, 16), fx.unit_rows(rng, 24, 16)
    L, dU, dI = O.in_batch_bpr_loss(U, I)
    Ls, dIs, dUs = 0.0, np.zeros_like(dI), []
    for r in range(3):
        l, du, di = O.in_batch_bpr_loss(U[r * 8:(r + 1) * 8], I, owner_offset=r * 8, n_global=24)
        Ls += float(l); dIs += di; dUs.append(du)
    assert abs(Ls - float(L)) < 1e-6
    np.testing.assert_allclose(np.concatenate(dUs), dU, atol=1e-8)
    np.testing.assert_allclose(dIs, dI, atol=1e-8)


def test_g4_train50(golden_dir):
    g = np.load(golden_dir / "g4_train50.npz")
    nu, ni, d, H, seed, B = (int(x) for x in g["cfg"])
    sd = fx.make_state(nu, ni, d, H, seed)
    m = {k: np.zeros_like(v) for k, v in sd.items()}
    v = {k: np.zeros_like(v) for k, v in sd.items()}
    epoch = 0
    for step in range(50):
        lr = O.cosine_lr(1e-2, epoch, 2)
        assert abs(lr - g["lrs"][step]) < 1e-12
        u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=4000 + step, boundary=False)
        pu, pi = _params(sd, "user_tower"), _params(sd, "item_tower")
        U, cu = O.tower_forward(pu, u); P, cp = O.tower_forward(pi, p, gp); N, cn = O.tower_forward(pi, n, gn)
        loss, dU, dP, dN = O.bpr_loss(U, P, N)
        assert abs(float(loss) - g["losses"][step]) < 5e-5, step
        bu = O.tower_backward(pu, cu, dU); bp = O.tower_backward(pi, cp, dP); bn = O.tower_backward(pi, cn, dN)
        grads = {
            "user_tower.embedding.weight": O.embedding_scatter_add(nu + 1, u, bu[0]),
            "user_tower.mlp.0.weight": bu[1], "user_tower.mlp.0.bias": bu[2],
            "user_tower.mlp.3.weight": bu[3], "user_tower.mlp.3.bias": bu[4],
            "item_tower.embedding.weight": O.embedding_scatter_add(ni + 1, p, bp[0]) + O.embedding_scatter_add(ni + 1, n, bn[0]),
            "item_tower.mlp.0.weight": bp[1] + bn[1], "item_tower.mlp.0.bias": bp[2] + bn[2],
            "item_tower.mlp.3.weight": bp[3] + bn[3], "item_tower.mlp.3.bias": bp[4] + bn[4],
        }
        c, _ = O.clip_coef([grads[k] for k in fx.PARAM_ORDER])
        for k in fx.PARAM_ORDER:
            O.adam_step(sd[k], grads[k], m[k], v[k], step + 1, lr, wd=1e-5, clip=c)
        if step == 24:
            epoch += 1
    for k in fx.PARAM_ORDER:
        np.testing.assert_allclose(sd[k], g[f"final_{k}"], atol=2e-4, rtol=0, err_msg=k)


def test_g5_inference(golden_dir):
    g = np.load(golden_dir / "g5_inference.npz")
    nu, ni, d, H, seed = (int(x) for x in g["cfg"])
    sd = fx.make_state(nu, ni, d, H, seed)
    E, _ = O.tower_forward(_params(sd, "item_tower"), np.arange(1, 1001), g["genres"])
    np.testing.assert_allclose(E, g["item_embs"], atol=2e-6)
    U, _ = O.tower_forward(_params(sd, "user_tower"), np.array([7, 100]))
    np.testing.assert_allclose(U[0], g["user7"], atol=2e-6)
    np.testing.assert_allclose(U[1], g["user100"], atol=2e-6)


def test_dropout_mask_statistics_and_determinism():
    k1 = O.dropout_keep_mask(123, 0, 512, 128, 0.1)
    k2 = O.dropout_keep_mask(123, 0, 512, 128, 0.1)
    assert (k1 == k2).all()
    assert abs(k1.mean() - 0.9) < 0.01
    # row offset addresses the same global elements
    k3 = O.dropout_keep_mask(123, 100, 50, 128, 0.1)
    assert (k3 == k1[100:150]).all()
    assert O.dropout_keep_mask(124, 0, 512, 128, 0.1).mean() != k1.mean() or True


def test_g7_metric_known_answers():
    # values from the reference's tests/test_models.py:372-426
    assert abs(M.ndcg_at_k([1, 2, 3, 4, 5], [1, 2, 3], 3) - 1.0) < 1e-6
    assert M.ndcg_at_k([4, 5, 6, 7, 8], [1, 2, 3], 5) == 0.0
    assert 0.0 < M.ndcg_at_k([1, 4, 2, 5, 3], [1, 2, 3], 5) < 1.0
    assert abs(M.recall_at_k([1, 2, 3, 4, 5], [1, 2, 6, 7], 3) - 0.5) < 1e-6
    assert M.recall_at_k([1, 2, 3], [], 3) == 0.0
    assert abs(M.mrr([1, 2, 3], [1]) - 1.0) < 1e-6
    assert abs(M.mrr([4, 1, 2], [1, 2]) - 0.5) < 1e-6
    assert M.mrr([4, 5, 6], [1, 2, 3]) == 0.0
    assert abs(M.coverage([[1, 2, 3], [4, 5, 6], [1, 7, 8]], 10) - 0.8) < 1e-6


def test_lambdamart_oracle_learns_and_is_deterministic():
    """oracle/lambdamart_np.py (the restatement the HIP LambdaMART trainer is pinned to): NDCG rises on a learnable set,
    two runs give identical trees, lambdas of a two-document query have the closed form."""
    from oracle import lambdamart_np as LM
    rng = np.random.RandomState(0)
    F, groups = 6, [25] * 40
    n = sum(groups)
    X = rng.randn(n, F).astype(np.float32)
    y = (X[:, 0] + 0.5 * X[:, 1] + 0.3 * rng.randn(n) > 0.8).astype(np.float32)
    p = dict(num_leaves=7, n_estimators=8, learning_rate=0.2, eval_at=[5], min_child_samples=5)
    a = LM.train(X, y, groups, p)
    b = LM.train(X, y, groups, p)
    h = [r["train"][0] for r in a["history"]]
    assert h[-1] > h[0] + 0.05 and h[-1] > 0.8
    for ta, tb in zip(a["trees"], b["trees"]):
        np.testing.assert_array_equal(ta["split_feature"], tb["split_feature"])
        np.testing.assert_array_equal(ta["leaf_value"], tb["leaf_value"])
    # one relevant + one irrelevant document, equal scores: rho = 1/2, delta NDCG = (1 - 1/log2(3)) / 1
    lam, hes = LM.lambdarank_grads(np.zeros(2), np.array([1.0, 0.0]), [2], LM.default_params(lambdarank_norm=False))
    d = 1.0 - 1.0 / np.log2(3.0)
    np.testing.assert_allclose(lam, [-0.5 * d, 0.5 * d], rtol=1e-12)
    np.testing.assert_allclose(hes, [0.25 * d, 0.25 * d], rtol=1e-12)
