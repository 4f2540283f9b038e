#!/usr/bin/env python3
"""BASELINE.json configs[1] shape: ML-1M tables, d=64, B=8192 in-batch negatives, dense Adam.  For rocprofv3."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from recommendit_amd import TwoTowerModel  # noqa: E402
from recommendit_amd.trainer import HipBPRTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda")
m = TwoTowerModel(6040, 3952, 64, 128, dropout=0.1); m.train()
tr = HipBPRTrainer(m, B, loss_mode="inbatch", table_opt="dense")
g = torch.Generator(device=dev); g.manual_seed(0)
u = torch.randint(1, 6041, (B,), device=dev, generator=g)
it = torch.randint(1, 3953, (B,), device=dev, generator=g)
gen = (torch.rand((B, 18), device=dev, generator=g) < 0.1).float()
for _ in range(5):
    tr.step(u, it, gen)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 100
for _ in range(n):
    tr.step(u, it, gen)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"B={B}: {dt * 1e6:.1f} us/step -> {B / dt / 1e6:.2f} M pairs/s")
