"""cfg5 serve leg alone (for rocprofv3): python tools/serve_bench.py [n_items]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench as B  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = B.make_model(1_000_000, N, B.D, B.H, seed=1234, user_seed=1000, item_seed=2000)
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, B.D), device=dev, generator=g)
X = (X / X.norm(dim=1, keepdim=True)).contiguous()
out = B.leg_serve(model, X, 1_000_000, dev, cpu=False)
print({k: v for k, v in out.items() if k != "ranker"})
