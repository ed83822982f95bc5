"""Micro-bench of the two backward kernels (B=8, 512x1024, 18 x 256^2, fp32, planar layout)."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib as L
lib = L.load()
B, H, W, P, N, nrows = 8, 512, 1024, 256, 18, 4
P_ = lambda t: ctypes.c_void_p(t.data_ptr())
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
LAYOUTS = ((L.LAYOUT_BNCHW, "planar"), (L.LAYOUT_BCHWN, "reference"))
if os.environ.get("LAYOUT"): LAYOUTS = tuple(l for l in LAYOUTS if l[1] == os.environ["LAYOUT"])
for lay, name in LAYOUTS:
    gp = torch.rand((B, N, 3, P, P) if lay == L.LAYOUT_BNCHW else (B, 3, P, P, N), device="cuda")
    ge = torch.empty((B, 3, H, W), device="cuda")
    t1 = timeit(lambda: lib.omni_equi2pers_bwd(P_(gp), P_(ge), 0, B, 3, H, W, P, P, nrows, ctypes.c_float(80), ctypes.c_float(80), lay, None))
    b1 = B * 3 * (H * W + P * P * N) * 4
    ge1 = torch.rand((B, 1, H, W), device="cuda")
    gp1 = torch.empty((B, N, 1, P, P) if lay == L.LAYOUT_BNCHW else (B, 1, P, P, N), device="cuda")
    t2 = timeit(lambda: lib.omni_pers2equi_bwd(P_(ge1), P_(gp1), 0, B, 1, P, P, H, W, nrows, ctypes.c_float(80), ctypes.c_float(80), lay, None))
    b2 = B * 1 * (H * W + P * P * N) * 4
    print(f"{name:9s}: equi2pers_bwd {t1*1e6:7.1f} us {b1/t1/1e9:6.0f} GB/s | pers2equi_bwd {t2*1e6:7.1f} us {b2/t2/1e9:6.0f} GB/s")
