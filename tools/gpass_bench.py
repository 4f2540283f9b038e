"""Stored-G in-batch pass vs the two-sweep form: equality + timing.  python tools/gpass_bench.py [B] [d]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendit_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
PREC = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = L.lib(); dev = L.device(); st = L.stream_ptr()
g = torch.Generator(device="cpu").manual_seed(1)
U = torch.nn.functional.normalize(torch.randn(B, d, generator=g), dim=1).to(dev)
I = torch.nn.functional.normalize(torch.randn(B, d, generator=g), dim=1).to(dev)
f32 = dict(dtype=torch.float32, device=dev)
pos = torch.empty(B, **f32); r = torch.empty(B, **f32); r2 = torch.empty(B, **f32)
dU = torch.empty(B, d, **f32); dI = torch.empty(B, d, **f32); dU2 = torch.empty(B, d, **f32); dI2 = torch.empty(B, d, **f32)
lp = torch.zeros(max(1024, lib.rihip_inbatch_workspace_doubles(B)), dtype=torch.float64, device=dev)
ws = torch.empty(lib.rihip_inbatch_workspace_floats(B, B, d), **f32)
gm = torch.empty(lib.rihip_inbatch_gmat_floats(B, B), **f32)
L.check(lib.rihip_rowdot(U.data_ptr(), I.data_ptr(), B, 0, d, pos.data_ptr(), st), "rowdot")

def two_sweep():
    L.check(lib.rihip_inbatch_sweep(1, U.data_ptr(), B, 0, I.data_ptr(), B, 0, d, pos.data_ptr(), None, B, dU.data_ptr(),
                                    r.data_ptr(), lp.data_ptr(), ws.data_ptr(), 0, st), "u")
    L.check(lib.rihip_inbatch_sweep(0, I.data_ptr(), B, 0, U.data_ptr(), B, 0, d, pos.data_ptr(), r.data_ptr(), B,
                                    dI.data_ptr(), None, None, ws.data_ptr(), 0, st), "i")
def upass():
    L.check(lib.rihip_inbatch_user_pass(U.data_ptr(), B, 0, I.data_ptr(), B, 0, d, pos.data_ptr(), B, dU2.data_ptr(),
                                        r2.data_ptr(), lp.data_ptr(), ws.data_ptr(), gm.data_ptr(), PREC, st), "up")
def ipass():
    L.check(lib.rihip_inbatch_item_pass(gm.data_ptr(), U.data_ptr(), B, 0, B, 0, d, r2.data_ptr(), B, dI2.data_ptr(),
                                        ws.data_ptr(), PREC, st), "ip")
two_sweep(); upass(); ipass(); torch.cuda.synchronize()
print("dU maxrel", float(((dU - dU2).abs() / (dU.abs() + 1e-12)).max()), "dU equal", torch.equal(dU, dU2), "r equal", torch.equal(r, r2),
      "dI maxabs", float((dI - dI2).abs().max()), "scale", float(dI.abs().max()))
def tm(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
fl = 2.0 * B * B * d
for name, fn, mult in (("two_sweep", two_sweep, 4), ("user_pass", upass, 2), ("item_pass", ipass, 1)):
    ms = tm(fn)
    print(f"{name}: {ms:.3f} ms  executed {mult * fl / ms / 1e9:.1f} TF/s")
