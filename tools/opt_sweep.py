"""Time the resample pair under a sweep of one library option (product build), interleaved, with a result check.

    python tools/opt_sweep.py e2p_store 0 1 2 3        (env B, P, H, W, NROWS, HALF, OP=e2p|p2e|both, ITERS)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi

opt, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
E = os.environ.get
B, P, H, W, nrows = int(E("B", "8")), int(E("P", "256")), int(E("H", "512")), int(E("W", "1024")), int(E("NROWS", "4"))
half, op, iters = E("HALF", "0") == "1", E("OP", "both"), int(E("ITERS", "30"))
N = {3: 10, 4: 18, 5: 26, 6: 46}[nrows]
dev, dt = "cuda:0", torch.float16 if half else torch.float32
lay = L.LAYOUT_BNCHW
erp = torch.rand((B, 3, H, W), device=dev).to(dt)
pp = torch.rand((B, N, 1, P, P), device=dev).to(dt)
s = 2 if half else 4
b1 = B * 3 * (H * W + P * P * N) * s
b2 = B * 1 * (P * P * N + H * W) * s


def timeit(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


f1 = lambda: equi2pers_patches(erp, 80, nrows, P, layout=lay)
f2 = lambda: pers2equi(pp, 80, nrows, P, (H, W), None, layout=lay)
ref1 = ref2 = None
best = {}
for rep in range(3):
    for v in vals:
        L.set_option(opt, v)
        line = f"{opt}={v:>3d} B={B} P={P} {H}x{W} nrows={nrows} {'f16' if half else 'f32'}:"
        if op in ("e2p", "both"):
            o = f1()
            if ref1 is None: ref1 = o.clone()
            ok = torch.equal(o, ref1)
            t = timeit(f1, iters); best[(v, 1)] = min(best.get((v, 1), 1e9), t)
            line += f" equi2pers {t*1e6:6.1f} us {b1/t/1e9:5.0f} GB/s {'same bits' if ok else 'DIFFERENT (max %.3g)' % (o.float() - ref1.float()).abs().max().item()} |"
        if op in ("p2e", "both"):
            o = f2()
            if ref2 is None: ref2 = o.clone()
            ok = torch.equal(o, ref2)
            t = timeit(f2, iters); best[(v, 2)] = min(best.get((v, 2), 1e9), t)
            line += f" pers2equi {t*1e6:6.1f} us {b2/t/1e9:5.0f} GB/s {'same bits' if ok else 'DIFFERENT (max %.3g)' % (o.float() - ref2.float()).abs().max().item()}"
        print(line, flush=True)
print("best of 3:", " ".join(f"[{opt}={v} " + " ".join(f"{'e2p' if k == 1 else 'p2e'} {best[(v, k)]*1e6:.1f}us" for k in (1, 2) if (v, k) in best) + "]" for v in vals))
