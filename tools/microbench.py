#!/usr/bin/env python3
"""Per-kernel timing of the training-step pieces (HIP events on the launch stream).  python tools/microbench.py [B]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from recommendit_amd import TwoTowerModel  # noqa: E402
from recommendit_amd.trainer import HipBPRTrainer  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    d, H = 128, 128
    dev = torch.device("cuda")
    m = TwoTowerModel(1_000_000, 1_000_000, d, H, dropout=0.1)
    m.train()
    tr = HipBPRTrainer(m, B, loss_mode="sampled", table_opt="sparse")
    g = torch.Generator(device=dev); g.manual_seed(0)
    u = torch.randint(1, 1_000_000, (B,), device=dev, generator=g)
    it = torch.randint(1, 1_000_000, (2 * B,), device=dev, generator=g)
    gen = (torch.rand((2 * B, 18), device=dev, generator=g) < 0.1).float()
    from recommendit_amd import _lib as L
    tr._st = L.stream_ptr()
    uk, ik = ["user_tower.mlp.0.weight", "user_tower.mlp.0.bias", "user_tower.mlp.3.weight", "user_tower.mlp.3.bias"], \
             ["item_tower.mlp.0.weight", "item_tower.mlp.0.bias", "item_tower.mlp.3.weight", "item_tower.mlp.3.bias"]
    fl_u = 2 * (d * H + H * d) * B
    fl_i = 2 * ((d + 18) * H + H * d) * 2 * B
    t = timeit(lambda: tr._fwd(tr.utab, u, None, uk, tr.U, tr.hidU, tr.denU, 1))
    print(f"fwd user  B={B}: {t:8.1f} us  {fl_u / t / 1e6:6.1f} TF/s")
    t = timeit(lambda: tr._fwd(tr.itab, it, gen, ik, tr.I, tr.hidI, tr.denI, 2))
    print(f"fwd item  B={2 * B}: {t:8.1f} us  {fl_i / t / 1e6:6.1f} TF/s")
    tr.dU.normal_(); tr.dI.normal_()
    t = timeit(lambda: tr._bwd(tr.utab, u, None, uk, tr.dU, tr.U, tr.denU, tr.hidU, tr.dXu))
    print(f"bwd user  B={B}: {t:8.1f} us  {2 * fl_u / t / 1e6:6.1f} TF/s")
    t = timeit(lambda: tr._bwd(tr.itab, it, gen, ik, tr.dI, tr.I, tr.denI, tr.hidI, tr.dXi))
    print(f"bwd item  B={2 * B}: {t:8.1f} us  {2 * fl_i / t / 1e6:6.1f} TF/s")
    pp = tr.part.data_ptr()
    t = timeit(lambda: tr.uopt.group_reduce(u, tr.dXu, pp, tr._st))
    print(f"rows group+reduce user: {t:8.1f} us")
    t = timeit(lambda: tr.iopt.group_reduce(it, tr.dXi, pp, tr._st))
    print(f"rows group+reduce item: {t:8.1f} us")
    t = timeit(lambda: tr.step(u, it, gen))
    print(f"full sampled step: {t:8.1f} us  -> {B / t:6.2f} M pairs/s")


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] == "small"):
    main()


def small_batch(B=1024, d=64, H=128, n=200):
    """launch-bound regime (the reference's default batch): eager vs hipGraph replay"""
    import time
    dev = torch.device("cuda")
    for use_graph in (False, True):
        m = TwoTowerModel(6040, 3952, d, H, dropout=0.1)
        m.train()
        tr = HipBPRTrainer(m, B, loss_mode="sampled", table_opt="dense", use_graph=use_graph)
        g = torch.Generator(device=dev); g.manual_seed(0)
        u = torch.randint(1, 6040, (B,), device=dev, generator=g)
        it = torch.randint(1, 3952, (2 * B,), device=dev, generator=g)
        gen = (torch.rand((2 * B, 18), device=dev, generator=g) < 0.1).float()
        for _ in range(5):
            tr.step(u, it, gen)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            tr.step(u, it, gen)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"B={B} d={d} dense-Adam sampled step, use_graph={use_graph}: {dt * 1e6:8.1f} us/step -> {B / dt / 1e6:6.2f} M pairs/s")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "small":
    small_batch(256); small_batch(1024); small_batch(8192)
