"""Phase times of the one-launch training step (csrc/step_persistent.hip): workgroup 0 stamps the 100 MHz wall clock at
every phase boundary into the last 8 scratch doubles.  python tools/persist_phases.py [B] [d] [H]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from recommendit_amd import TwoTowerModel
from recommendit_amd.trainer import HipBPRTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
torch.manual_seed(0)
m = TwoTowerModel(6040, 3952, embed_dim=d, hidden_dim=H, dropout=0.1); m.train()
tr = HipBPRTrainer(m, B, loss_mode="sampled", table_opt="dense", persistent=True)
g = torch.Generator(device="cuda"); g.manual_seed(1)
u = torch.randint(1, 6041, (B,), device="cuda", generator=g)
it = torch.randint(1, 3953, (2 * B,), device="cuda", generator=g)
gg = (torch.rand((2 * B, 18), device="cuda", generator=g) < 0.1).float()
names = ["A fwd", "barrier0", "B loss+bwd", "barrier1", "C wgrad+scatter", "barrier2", "D clip+adam"]
acc = np.zeros(7)
n = 200
for i in range(n + 20):
    tr.step(u, it, gg)
    if i >= 20:
        torch.cuda.synchronize()
        st = tr._pscratch[-8:].cpu().numpy()
        acc += np.diff(st) / 100.0      # us
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(500):
    tr.step(u, it, gg)
torch.cuda.synchronize()
print(f"B={B} d={d} H={H}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us/step (back-to-back)")
for nm, v in zip(names, acc / n):
    print(f"  {nm:18s} {v:7.2f} us")
print(f"  sum (workgroup 0)  {acc.sum() / n:7.2f} us")
tr.check_errors()
