#!/usr/bin/env python3
"""G9: run the REFERENCE training loop (src/training/train_embeddings.py, unmodified, imported in this
container only) on the seeded ML-1M-shaped synthetic set and record its loss curve and the
retrieval-only NDCG under the run_evaluate protocol (first 200 test users) AND over all test users (same metric,
30x less sampling noise), for several seeds -> tests/golden/g9_reference_ndcg.json.

The only shim is an in-process stand-in for the missing `pydantic_settings` package, so that
`src.config` imports; it carries no logic.  Usage: python oracle/make_golden_g9.py [n_seeds] [epochs] [first_seed] [out_json]   (shards are merged by hand)
"""
import json
import sys
import tempfile
import time
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")

import pydantic  # noqa: E402

stub = types.ModuleType("pydantic_settings")


class BaseSettings(pydantic.BaseModel):
    model_config = pydantic.ConfigDict(extra="ignore")


stub.BaseSettings = BaseSettings
stub.SettingsConfigDict = dict
sys.modules["pydantic_settings"] = stub
sys.path.insert(0, str(REF))

from recommendit_amd.synthetic import ml1m_like, write_ml1m_files  # noqa: E402
from oracle import metrics_np as M  # noqa: E402
from oracle import retrieval_np as R  # noqa: E402


def evaluate(model, ratings, movies, genre_dict, max_users=200):
    """retrieval-only form of src/pipelines/run_pipeline.py:153-230 (exact IP, default features => ranker ties)."""
    n_users = ratings["user_id"].nunique()
    n_test = max(1, int(len(ratings) * 0.1 / n_users))
    test = ratings.sort_values("timestamp").groupby("user_id").tail(n_test)
    eval_users = test["user_id"].unique()
    if max_users:
        eval_users = eval_users[:max_users]
    item_ids = sorted(movies["item_id"].unique().tolist())
    gm = np.stack([genre_dict.get(i, np.zeros(18, np.float32)) for i in item_ids])
    E = R.normalize_rows(model.get_item_embeddings(item_ids, gm))
    U = R.normalize_rows(np.stack([model.get_user_embedding(int(u)) for u in eval_users]))
    _, rows = R.topk_ip_exact(U, E, 500)
    ids = np.asarray(item_ids)[rows]
    truth = {int(u): g[g["rating"] >= 4]["item_id"].tolist() for u, g in test[test["user_id"].isin(eval_users)].groupby("user_id")}
    recs = {int(u): [int(x) for x in ids[i][:20]] for i, u in enumerate(eval_users)}
    return {f"ndcg@{k}": M.mean_ndcg(recs, truth, k) for k in (5, 10, 20)}


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    first_seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    out_path = Path(sys.argv[4]) if len(sys.argv) > 4 else ROOT / "tests" / "golden" / "g9_reference_ndcg.json"
    import logging
    logging.basicConfig(level=logging.INFO)
    from src.training.train_embeddings import EmbeddingTrainer
    ratings, movies, gm = ml1m_like(seed=0)
    out = {"data": "recommendit_amd.synthetic.ml1m_like(seed=0)", "epochs": epochs, "batch_size": 1024, "lr": 1e-3,
           "embed_dim": 64, "runs": []}
    with tempfile.TemporaryDirectory() as td:
        write_ml1m_files(td, ratings, movies, 6040)
        for seed in range(first_seed, first_seed + n_seeds):
            torch.manual_seed(seed)
            np.random.seed(seed)
            t0 = time.time()
            tr = EmbeddingTrainer(data_dir=td, model_output_path=str(Path(td) / f"tt{seed}.pt"), embed_dim=64,
                                  epochs=epochs, batch_size=1024, learning_rate=1e-3, device="cpu")
            # capture the per-epoch losses the reference logs
            losses = []

            class H(logging.Handler):
                def emit(self, rec):
                    msg = rec.getMessage()
                    if msg.startswith("Epoch") and "loss:" in msg:
                        losses.append(float(msg.split("loss:")[1].split()[0]))
            h = H()
            logging.getLogger("src.training.train_embeddings").addHandler(h)
            model = tr.train()
            logging.getLogger("src.training.train_embeddings").removeHandler(h)
            gd = tr._build_item_genre_dict(movies)
            res = evaluate(model, ratings, movies, gd)
            res.update({k + "_all_users": v for k, v in evaluate(model, ratings, movies, gd, max_users=None).items()})
            res.update(seed=seed, epoch_losses=losses, seconds=time.time() - t0)
            print(json.dumps(res), flush=True)
            out["runs"].append(res)
            out_path.write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
