#!/usr/bin/env python3
"""Tower kernel time vs rows (intercept = per-launch prologue, slope = per-row cost).  python tools/tower_scale.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from recommendit_amd import TwoTowerModel  # noqa: E402
from recommendit_amd import _lib as L  # noqa: E402
from recommendit_amd.trainer import HipBPRTrainer  # noqa: E402
from tools.microbench import timeit  # noqa: E402

d, H = 128, 128
dev = torch.device("cuda")
uk = ["user_tower.mlp.0.weight", "user_tower.mlp.0.bias", "user_tower.mlp.3.weight", "user_tower.mlp.3.bias"]
ik = ["item_tower.mlp.0.weight", "item_tower.mlp.0.bias", "item_tower.mlp.3.weight", "item_tower.mlp.3.bias"]
m = TwoTowerModel(1_000_000, 1_000_000, d, H, dropout=0.1)
m.train()
for B in (4096, 16384, 65536, 131072, 262144):
    tr = HipBPRTrainer(m, B, loss_mode="sampled", table_opt="sparse")
    g = torch.Generator(device=dev); g.manual_seed(0)
    u = torch.randint(1, 1_000_000, (B,), device=dev, generator=g)
    it = torch.randint(1, 1_000_000, (2 * B,), device=dev, generator=g)
    gen = (torch.rand((2 * B, 18), device=dev, generator=g) < 0.1).float()
    tr._st = L.stream_ptr()
    tr.dU.normal_(); tr.dI.normal_()
    fu = timeit(lambda: tr._fwd(tr.utab, u, None, uk, tr.U, tr.hidU, tr.denU, 1))
    fi = timeit(lambda: tr._fwd(tr.itab, it, gen, ik, tr.I, tr.hidI, tr.denI, 2))
    bu = timeit(lambda: tr._bwd(tr.utab, u, None, uk, tr.dU, tr.U, tr.denU, tr.hidU, tr.dXu))
    bi = timeit(lambda: tr._bwd(tr.itab, it, gen, ik, tr.dI, tr.I, tr.denI, tr.hidI, tr.dXi))
    print(f"B={B:7d}: fwd user {fu:7.1f} us ({B} rows)  fwd item {fi:7.1f} us ({2 * B} rows)  bwd user {bu:7.1f}  bwd item {bi:7.1f}")
    del tr
