"""Batched serve loop timed three ways (device-tensor ids, windows of 12 / 48 batches, ids as a host list)."""
import os, sys, time, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from recommendit_amd import synthetic as GB, FAISSIndex, LightGBMRanker, TwoTowerModel
from recommendit_amd.recommender import GpuFeatureStore, GpuRecommendationPipeline, feature_columns
dev = torch.device("cuda", 0)
N, nu = 1_000_000, 100_000
torch.manual_seed(0)
model = TwoTowerModel(nu, N, embed_dim=128, hidden_dim=128); model.eval()
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, 128), device=dev, generator=g); X = (X / X.norm(dim=1, keepdim=True)).contiguous()
ivf = FAISSIndex(embed_dim=128, n_lists=100, n_probe=10); ivf.build_from_device(X, np.arange(1, N + 1))
forest = GB.random_forest_model(500, 63, 50, seed=4, names=feature_columns())
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "f.lgbm"); open(p, "w").write(GB.write_text_model(forest)); ranker = LightGBMRanker.load(p)
store = GpuFeatureStore(8, 8)
store._dev = (torch.rand((nu + 1, 24), device=dev, generator=g, dtype=torch.float64), torch.rand((N + 1, 23), device=dev, generator=g, dtype=torch.float64))
pipe = GpuRecommendationPipeline(model, ivf, ranker, store, top_k_candidates=500, top_k_results=20)
nq = 256
uids = [torch.randint(1, nu + 1, (nq,), device=dev, generator=g) for _ in range(3)]
lists = [u.tolist() for u in uids]
for _ in range(40):
    pipe.recommend_batch(uids[0])
torch.cuda.synchronize()
for name, ids, n in (("device ids, 12", uids, 12), ("device ids, 48", uids, 48), ("host lists, 48", lists, 48), ("device ids, 200", uids, 200)):
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            pipe.recommend_batch(ids[i % 3])
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    ts.sort()
    t0 = time.perf_counter()
    for i in range(n):
        pipe.recommend_batch(ids[i % 3])
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    print(f"{name}: median {ts[2]:.3f} ms/batch, min {ts[0]:.3f}; host time per call {host:.3f} ms")
