"""Kernel micro-bench for the resample pair (used under rocprofv3 for PMC passes)."""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8); ap.add_argument("--P", type=int, default=256)
ap.add_argument("--H", type=int, default=512); ap.add_argument("--W", type=int, default=1024)
ap.add_argument("--nrows", type=int, default=4); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--layout", default="planar"); ap.add_argument("--half", action="store_true")
ap.add_argument("--copy", action="store_true", help="also time a torch device copy of the same byte count")
a = ap.parse_args()
N = {3: 10, 4: 18, 5: 26, 6: 46}[a.nrows]
dev = "cuda:0"; dt = torch.float16 if a.half else torch.float32
lay = L.LAYOUT_BNCHW if a.layout == "planar" else L.LAYOUT_BCHWN
erp = torch.rand((a.B, 3, a.H, a.W), device=dev).to(dt)
pp = (torch.rand((a.B, N, 1, a.P, a.P), device=dev) if lay == L.LAYOUT_BNCHW else torch.rand((a.B, 1, a.P, a.P, N), device=dev)).to(dt)

def timeit(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

s = 2 if a.half else 4
b1 = a.B * 3 * (a.H * a.W + a.P * a.P * N) * s
b2 = a.B * 1 * (a.P * a.P * N + a.H * a.W) * s
t1 = timeit(lambda: equi2pers_patches(erp, 80, a.nrows, a.P, layout=lay), a.iters)
t2 = timeit(lambda: pers2equi(pp, 80, a.nrows, a.P, (a.H, a.W), None, layout=lay), a.iters)
print(f"B={a.B} P={a.P} {a.H}x{a.W} nrows={a.nrows} {a.layout} {'f16' if a.half else 'f32'}: "
      f"equi2pers {t1*1e6:.1f} us {b1/t1/1e9:.0f} GB/s | pers2equi {t2*1e6:.1f} us {b2/t2/1e9:.0f} GB/s | "
      f"pair {(b1+b2)/(t1+t2)/1e9:.0f} GB/s ({(b1+b2)/(t1+t2)/8e12*100:.1f}% of 8 TB/s)")
if a.copy:
    src = torch.empty(b1 // 8, device=dev, dtype=torch.float32); dst = torch.empty_like(src)
    t3 = timeit(lambda: dst.copy_(src), a.iters)
    print(f"device copy of {b1/1e6:.0f} MB total traffic: {t3*1e6:.1f} us {b1/t3/1e9:.0f} GB/s")
