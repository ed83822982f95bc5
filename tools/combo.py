"""Time one resample operator under several option COMBINATIONS (product build), interleaved, best of REPS, with a result check.

    OP=e2p python tools/combo.py "" "e2p_late=1" "e2p_late=1,e2p_store=1"        (env B, P, H, W, NROWS, HALF, ITERS, REPS)
The geometry cache is cleared at every switch (some options act when a handle is built).
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi

E = os.environ.get
B, P, H, W, nrows = int(E("B", "8")), int(E("P", "256")), int(E("H", "512")), int(E("W", "1024")), int(E("NROWS", "4"))
half, op, iters, reps = E("HALF", "0") == "1", E("OP", "e2p"), int(E("ITERS", "30")), int(E("REPS", "3"))
N = {3: 10, 4: 18, 5: 26, 6: 46}[nrows]
dev, dt = "cuda:0", torch.float16 if half else torch.float32
lay = L.LAYOUT_BNCHW
erp = torch.rand((B, 3, H, W), device=dev).to(dt)
pp = torch.rand((B, N, 1, P, P), device=dev).to(dt)
s = 2 if half else 4
nbytes = B * 3 * (H * W + P * P * N) * s if op == "e2p" else B * 1 * (P * P * N + H * W) * s
fn = (lambda: equi2pers_patches(erp, 80, nrows, P, layout=lay)) if op == "e2p" else (lambda: pers2equi(pp, 80, nrows, P, (H, W), None, layout=lay))
lib = L.load()


def timeit(n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


combos = [dict(kv.split("=") for kv in c.split(",") if kv) for c in sys.argv[1:]] or [{}]
allkeys = sorted({k for c in combos for k in c})
defaults = {}
for k in allkeys:
    import ctypes
    v = ctypes.c_int(0); L.check(lib.omni_get_option(k.encode(), ctypes.byref(v)), "get_option"); defaults[k] = v.value
ref, best, same = None, {}, {}
for rep in range(reps):
    for i, c in enumerate(combos):
        for k in allkeys: L.set_option(k, int(c.get(k, defaults[k])))
        lib.omni_geometry_cache_clear()
        o = fn()
        if ref is None: ref = o.clone()
        same[i] = torch.equal(o, ref) if same.get(i, True) else False
        if not same[i] and rep == 0:
            print(f"   [{sys.argv[1 + i] if len(sys.argv) > 1 else ''}] DIFFERENT: max |d| {(o.float() - ref.float()).abs().max().item():.3g}")
        t = timeit(iters)
        best[i] = min(best.get(i, 1e9), t)
print(f"{op} B={B} P={P} {H}x{W} nrows={nrows} {'f16' if half else 'f32'} (best of {reps} x {iters} launches):")
for i, c in enumerate(combos):
    name = ",".join(f"{k}={v}" for k, v in c.items()) or "(defaults)"
    print(f"   {name:50s} {best[i]*1e6:7.1f} us {nbytes/best[i]/1e9:6.0f} GB/s  {'same bits' if same[i] else 'DIFFERENT'}")
