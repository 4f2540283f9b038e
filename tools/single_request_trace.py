"""Kernel timeline of ONE served request (cfg5 chain): python tools/single_request_trace.py  (run under
rocprofv3 --kernel-trace; tools/single_request_trace.sh prints the last request's kernels with durations and gaps)."""
import os, sys, tempfile
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
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1       # requests per batch
for i in range(30):
    pipe.recommend_batch([1 + (i * nb + j) % nu for j in range(nb)], graph=(mode == "graph"))
torch.cuda.synchronize()
if os.environ.get("RIHIP_FIN_PROBE"):
    import ctypes as C
    from recommendit_amd import _lib as L
    buf = (C.c_ulonglong * 16)()
    fn = L.lib().rihip_debug_fin_probe
    fn.argtypes = [C.c_void_p]
    fn(buf)
    v = list(buf)
    print("finalize (last launch, workgroup 0) stamps, us since start:", [round((x - v[0]) / 100.0, 1) if x >= v[0] else None for x in v[:12]])
