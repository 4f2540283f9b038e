"""LambdaMART forward alone: 500 trees x 63 leaves x 50 features on n candidates.  RIHIP_GBDT_WALK=8|4 forces a kernel family."""
import os, sys, time, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench as B
from recommendit_amd import synthetic as GB, LightGBMRanker
from recommendit_amd.recommender import feature_columns
forest = GB.random_forest_model(500, 63, 50, seed=4, names=feature_columns())
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "f.lgbm"); open(p, "w").write(GB.write_text_model(forest)); rk = LightGBMRanker.load(p)
g = torch.Generator(device="cuda"); g.manual_seed(0)
for n in (128000, 512000):
    X = torch.rand((n, 50), device="cuda", generator=g)
    med, best, _ = B.timed_blocks(lambda i: rk.predict_device(X), 10)
    print(f"walk={os.environ.get('RIHIP_GBDT_WALK', 'auto')} n={n}: {med * 1e3:.3f} ms = {n / med / 1e6:.0f} M candidates/s", flush=True)
