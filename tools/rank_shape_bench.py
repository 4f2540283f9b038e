"""Per-rank shapes of the multi-GPU in-batch step on ONE GPU: local users x all items.  python tools/rank_shape_bench.py [world]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendit_amd import _lib as L

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PREC = int(sys.argv[2]) if len(sys.argv) > 2 else 0
G, d = 65536, 128
B = G // W
off = B  # as if rank 1
lib = L.lib(); dev = L.device(); st = L.stream_ptr()
g = torch.Generator(device="cpu").manual_seed(1)
U = torch.nn.functional.normalize(torch.randn(B, d, generator=g), dim=1).to(dev)
I = torch.nn.functional.normalize(torch.randn(G, d, generator=g), dim=1).to(dev)
f32 = dict(dtype=torch.float32, device=dev)
pos = torch.empty(B, **f32); r = torch.empty(B, **f32)
dU = torch.empty(B, d, **f32); dI = torch.empty(G, d, **f32)
lp = torch.zeros(max(1024, lib.rihip_inbatch_workspace_doubles(B)), dtype=torch.float64, device=dev)
ws = torch.empty(max(lib.rihip_inbatch_workspace_floats(B, G, d), lib.rihip_inbatch_workspace_floats(G, B, d)), **f32)
gm = torch.empty(lib.rihip_inbatch_gmat_floats(B, G), **f32)
L.check(lib.rihip_rowdot(U.data_ptr(), I.data_ptr(), B, off, d, pos.data_ptr(), st), "rowdot")
def upass():
    L.check(lib.rihip_inbatch_user_pass(U.data_ptr(), B, off, I.data_ptr(), G, 0, d, pos.data_ptr(), G, dU.data_ptr(),
                                        r.data_ptr(), lp.data_ptr(), ws.data_ptr(), gm.data_ptr(), PREC, st), "up")
def ipass():
    L.check(lib.rihip_inbatch_item_pass(gm.data_ptr(), U.data_ptr(), B, off, G, 0, d, r.data_ptr(), G, dI.data_ptr(),
                                        ws.data_ptr(), PREC, st), "ip")
def tm(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
fl = 2.0 * B * G * d
tu, ti = tm(upass), tm(ipass)
print(f"world={W} B_local={B}: user pass {tu:.3f} ms = {2 * fl / tu / 1e9:.1f} TF/s, item pass {ti:.3f} ms = {fl / ti / 1e9:.1f} TF/s, "
      f"sum x{W} = {(tu + ti) * W:.2f} ms-equivalent (N=1: 25.7)")
