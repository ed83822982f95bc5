"""LDS box path vs gather path of equi2pers (planar layout; expected bit-exact), vs the C oracle, and timing at BASELINE shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from omnifusion_amd import _lib as L
from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
from oracle import c_oracle as co
dev = "cuda:0"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
N_ = {3: 10, 4: 18, 5: 26, 6: 46}
cfgs = [(2, 3, 512, 1024, 4, 256, torch.float32), (8, 3, 512, 1024, 4, 256, torch.float32), (8, 3, 512, 1024, 4, 128, torch.float32),
        (1, 3, 512, 1024, 4, 128, torch.float32), (1, 3, 1024, 2048, 6, 256, torch.float32), (8, 1, 512, 1024, 4, 32, torch.float32),
        (1, 3, 2048, 4096, 6, 512, torch.float16), (1, 3, 2048, 4096, 6, 512, torch.float32), (4, 3, 2048, 4096, 6, 512, torch.float16),
        (2, 3, 256, 512, 5, 64, torch.float16), (3, 2, 200, 336, 3, 64, torch.float32)]
for (B, C, H, W, nrows, P, dt) in cfgs:
    N = N_[nrows]
    x = torch.rand((B, C, H, W), device=dev).to(dt)
    L.set_option("e2p_gather", 0)
    a = equi2pers_patches(x, 80, nrows, P, layout=L.LAYOUT_BNCHW)
    ta = timeit(lambda: equi2pers_patches(x, 80, nrows, P, layout=L.LAYOUT_BNCHW))
    L.set_option("e2p_gather", 1)
    b = equi2pers_patches(x, 80, nrows, P, layout=L.LAYOUT_BNCHW)
    tb = timeit(lambda: equi2pers_patches(x, 80, nrows, P, layout=L.LAYOUT_BNCHW))
    L.set_option("e2p_gather", 0)
    byts = B * C * (P * P * N + H * W) * x.element_size()
    print(f"B={B} C={C} {H}x{W} nrows={nrows} P={P} {str(dt)[6:]}: equal={torch.equal(a, b)} maxdiff={(a.float()-b.float()).abs().max().item():.2e} "
          f"box {ta:.1f} us ({byts/ta/1e3:.0f} GB/s)  gather {tb:.1f} us ({byts/tb/1e3:.0f} GB/s)", flush=True)
    if H * W <= 512 * 1024 and dt == torch.float32 and B <= 2:
        ref, _, _, _ = co.equi2pers(x.cpu().numpy(), 80, nrows, P)
        d = np.abs(a.permute(0, 2, 3, 4, 1).cpu().numpy() - ref)
        print(f"     vs oracle (noise input): max {d.max():.2e}  frac>1e-3 {(d > 1e-3).mean():.1e}")
x = torch.rand((8, 3, 512, 1024), device=dev)
for nb in (1, 2, 4):
    L.set_option("e2p_nbuf", nb)
    t = timeit(lambda: equi2pers_patches(x, 80, 4, 256, layout=L.LAYOUT_BNCHW))
    print(f"  B=8 P=256 slots {nb}: {t:.1f} us ({8*3*(256*256*18+512*1024)*4/t/1e3:.0f} GB/s)")
L.set_option("e2p_nbuf", 0)
