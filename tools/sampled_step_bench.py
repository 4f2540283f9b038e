#!/usr/bin/env python3
"""Sampled-negative BPR step at B=65536 on 10M x 1M tables (bench.py's sampled leg) in a loop -- for rocprofv3."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from recommendit_amd import TwoTowerModel  # noqa: E402
from recommendit_amd.trainer import HipBPRTrainer  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nu, ni = 2_000_000, 1_000_000
dev = torch.device("cuda")
m = TwoTowerModel(nu, ni, 128, 128, dropout=0.1); m.train()
USE_GRAPH = len(sys.argv) > 2 and sys.argv[2] == "graph"
tr = HipBPRTrainer(m, B, loss_mode="sampled", table_opt="sparse", use_graph=USE_GRAPH)
g = torch.Generator(device=dev); g.manual_seed(0)
u = torch.randint(1, nu + 1, (B,), device=dev, generator=g)
it = torch.randint(1, ni + 1, (2 * B,), device=dev, generator=g)
gen = (torch.rand((2 * B, 18), device=dev, generator=g) < 0.1).float()
for _ in range(3):
    tr.step(u, it, gen)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
e0.record()
h0 = time.perf_counter()
for _ in range(20):
    tr.step(u, it, gen)
h1 = time.perf_counter()
e1.record(); torch.cuda.synchronize()
print(f"B={B} graph={USE_GRAPH}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us/step (host enqueue {(h1 - h0) / 20 * 1e6:.1f} us/step)")
