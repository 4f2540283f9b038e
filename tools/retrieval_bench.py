#!/usr/bin/env python3
"""Retrieval-only timing (for rocprofv3): python tools/retrieval_bench.py [nq] [iters]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os  # noqa: E402

if os.environ.get("RIHIP_LIB"):   # experiments: another build of the library (ablation variants of one kernel)
    from recommendit_amd import _lib as _L  # noqa: E402
    _L.LIB_PATH = Path(os.environ["RIHIP_LIB"]).resolve()
from recommendit_amd import FAISSIndex  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N, D = 1_000_000, 128
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.randn((N, D), device="cuda", generator=g); X = (X / X.norm(dim=1, keepdim=True)).contiguous()
idx = FAISSIndex(embed_dim=D, exact=True); idx.build_from_device(X, np.arange(N))
Q = torch.randn((nq, D), device="cuda", generator=g); Q = (Q / Q.norm(dim=1, keepdim=True)).contiguous()
idx.batch_search_device(Q, k=500, normalized=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters):
    idx.batch_search_device(Q, k=500, normalized=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
print(f"nq={nq}: {dt * 1e3:.3f} ms/batch -> {nq / dt:,.0f} q/s")
