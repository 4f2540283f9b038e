"""GPU: NDCG@10 parity protocol.  The reference training loop (unmodified
src/training/train_embeddings.py, run on CPU in the build container by oracle/make_golden_g9.py) and the HIP
trainer are both trained on the same seeded ML-1M-shaped synthetic set with the reference's settings (d=64,
B=1024, lr=1e-3, 10 epochs, dropout 0.1, one sampled negative per positive, dense Adam + L2 + clip) and scored
with the reference's run_evaluate protocol in its retrieval-only form.  Both procedures are stochastic
(different RNG streams for shuffling, negatives and dropout), so the comparison is between 6-seed means, and the
tolerance is the one BASELINE.json's north_star states, asserted outright: |mean NDCG@10 - reference mean| <= 0.002
* under the reference's own protocol (first 200 test users), and
* over ALL test users (same metric, no user-sampling noise).
Measured in round 1: 0.0002 and 0.0006.  The synthetic set makes NDCG@10 small (~0.02: with 10 epochs at lr 1e-3 the
model mostly learns popularity), so the per-epoch training-loss curve (every epoch mean within 4e-3 of the
reference's; measured <= 3e-4) is the tight check."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ndcg_and_loss_curve_match_reference_band(golden_dir, tmp_path):
    from recommendit_amd.synthetic import ml1m_like
    from recommendit_amd.train_embeddings import EmbeddingTrainer, retrieval_ndcg
    ref = json.loads((golden_dir / "g9_reference_ndcg.json").read_text())
    runs = ref["runs"]
    assert len(runs) >= 2 and ref["epochs"] == 10 and ref["batch_size"] == 1024
    ref_ndcg = np.array([r["ndcg@10"] for r in runs])
    ref_all = np.array([r["ndcg@10_all_users"] for r in runs])
    ref_loss = np.array([r["epoch_losses"] for r in runs])
    ratings, movies, _ = ml1m_like(seed=0)
    got_ndcg, got_all, got_loss = [], [], []
    for seed in range(len(runs)):
        tr = EmbeddingTrainer(model_output_path=str(tmp_path / f"tt{seed}.pt"), embed_dim=64, epochs=10, batch_size=1024,
                              learning_rate=1e-3, loss_mode="sampled", table_opt="dense", dropout=0.1, seed=seed)
        model = tr.train(ratings, movies)
        got_loss.append([h["loss"] for h in tr.history])
        got_ndcg.append(retrieval_ndcg(model, ratings, movies)["ndcg@10"])
        got_all.append(retrieval_ndcg(model, ratings, movies, n_eval_users=None)["ndcg@10"])
    got_ndcg, got_all, got_loss = np.array(got_ndcg), np.array(got_all), np.array(got_loss)
    print("reference NDCG@10:", ref_ndcg, "HIP NDCG@10:", got_ndcg)
    print("all users: reference", ref_all, ref_all.mean(), "HIP", got_all, got_all.mean())
    print("reference loss:", ref_loss.mean(0).round(4), "HIP loss:", got_loss.mean(0).round(4))
    # training loss curve: every epoch mean within 0.004 of the reference's epoch mean
    np.testing.assert_allclose(got_loss.mean(0), ref_loss.mean(0), atol=4e-3, rtol=0)
    np.testing.assert_allclose(got_loss.mean(0), ref_loss.mean(0), atol=1e-3, rtol=0)   # measured: <= 3e-4
    print("200-user protocol: |delta mean| =", abs(got_ndcg.mean() - ref_ndcg.mean()))
    print("all users:         |delta mean| =", abs(got_all.mean() - ref_all.mean()))
    assert abs(got_ndcg.mean() - ref_ndcg.mean()) <= 0.002, (got_ndcg, ref_ndcg)
    assert abs(got_all.mean() - ref_all.mean()) <= 0.002, (got_all, ref_all)
