#!/usr/bin/env python3
"""Phase stamps of the large-batch filter (diagnostic build with -DRIHIP_WIDE_PROBE): cycles between the stamps of one
stage of workgroup 0, for wave 0 (first half) and wave 4 (second half).  RIHIP_LIB=<probe build> python tools/wide_probe.py"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from recommendit_amd import _lib as L  # noqa: E402

if os.environ.get("RIHIP_LIB"):
    L.LIB_PATH = Path(os.environ["RIHIP_LIB"]).resolve()
from recommendit_amd import FAISSIndex  # noqa: E402

N, D, nq = 1_000_000, 128, 4096
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.randn((N, D), device="cuda", generator=g); X = (X / X.norm(dim=1, keepdim=True)).contiguous()
idx = FAISSIndex(embed_dim=D, exact=True); idx.build_from_device(X, np.arange(N))
Q = torch.randn((nq, D), device="cuda", generator=g); Q = (Q / Q.norm(dim=1, keepdim=True)).contiguous()
for _ in range(3):
    idx.batch_search_device(Q, k=500, normalized=True)
torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
fn = L.lib().rihip_debug_wide_probe
fn.argtypes = [C.POINTER(C.c_ulonglong)]; fn.restype = C.c_int
assert fn(out) == 0
names = ["top->mfma0 end", "barrier", "emit0", "flush", "barrier", "mfma1 (+dma wait)", "barrier", "emit1", "flush", "barrier"]
for half in (0, 1):
    st = [out[half * 16 + k] for k in range(11)]
    print(f"wave {half * 4}: " + ", ".join(f"{n} {st[k + 1] - st[k]}" for k, n in enumerate(names)) + f"  | stage total {st[10] - st[0]} (s_memtime ticks)")

inner = ["A reads", "B maxima+ballots", "C test+atomics", "D dump", "E stores"]
for half in (0, 1):
    st = [out[32 + half * 16 + k] for k in range(6)]
    print(f"wave {half * 4} survivor phase: " + ", ".join(f"{n} {st[k + 1] - st[k]}" for k, n in enumerate(inner)))
