#!/usr/bin/env python3
"""LambdaMART training throughput at the ML-1M ranker shape (6040 queries x ~480 documents, 50 features, 63 leaves):
python tools/lambdamart_bench.py [n_trees]"""
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from recommendit_amd import LightGBMRanker  # noqa: E402

n_trees = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(0)
nq, F = 6040, 50
sizes = np.clip(rng.lognormal(5.6, 0.9, nq).astype(int), 20, 9000)
n = int(sizes.sum())
X = rng.randn(n, F).astype(np.float32)
w = rng.randn(F)
y = ((X @ w + 2.0 * rng.randn(n)) > 5.0).astype(np.float32)
df = pd.DataFrame(X, columns=[f"f{i}" for i in range(F)])
df["label"] = y
df["query_id"] = np.repeat(np.arange(nq), sizes)
rk = LightGBMRanker(num_leaves=63, n_estimators=n_trees, learning_rate=0.05)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = rk.train(df, [f"f{i}" for i in range(F)], backend="hip")
dt = time.perf_counter() - t0
print(f"{n} rows x {F} features, {nq} queries, 63 leaves: {n_trees} trees in {dt:.2f} s = {dt / n_trees * 1e3:.1f} ms/tree "
      f"(incl. binning + upload); train ndcg@10 {res['train']['ndcg@10'][0]:.4f} -> {res['train']['ndcg@10'][-1]:.4f}")
