"""LDS path vs gather path of pers2equi (bit-exact), vs the C oracle, and timing at BASELINE shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from omnifusion_amd import _lib as L
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi, pers2equi_conf
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
for (B, C, H, W, nrows, P, dt) in [(2, 1, 512, 1024, 4, 256, torch.float32), (8, 1, 512, 1024, 4, 256, torch.float32), (8, 1, 512, 1024, 4, 128, torch.float32),
                                   (3, 3, 250, 500, 5, 64, torch.float32), (1, 2, 1024, 2048, 6, 256, torch.float32), (1, 1, 2048, 4096, 6, 512, torch.float16),
                                   (1, 1, 2048, 4096, 6, 512, torch.float32), (1, 1, 100, 333, 3, 32, torch.float32), (16, 1, 512, 1024, 4, 256, torch.float32)]:
    N = N_[nrows]
    x = torch.rand((B, N, C, P, P), device=dev).to(dt)
    L.set_option("p2e_gather", 0)
    a = pers2equi(x, 80, nrows, P, (H, W), None, layout=L.LAYOUT_BNCHW)
    ta = timeit(lambda: pers2equi(x, 80, nrows, P, (H, W), None, layout=L.LAYOUT_BNCHW))
    L.set_option("p2e_gather", 1)
    b = pers2equi(x, 80, nrows, P, (H, W), None, layout=L.LAYOUT_BNCHW)
    tb = timeit(lambda: pers2equi(x, 80, nrows, P, (H, W), None, layout=L.LAYOUT_BNCHW))
    L.set_option("p2e_gather", 0)
    byts = B * C * (P * P * N + H * W) * x.element_size()
    print(f"B={B} C={C} {H}x{W} nrows={nrows} P={P} {str(dt)[6:]}: equal={torch.equal(a, b)} maxdiff={(a.float()-b.float()).abs().max().item():.2e} "
          f"lds {ta:.1f} us ({byts/ta/1e3:.0f} GB/s)  gather {tb:.1f} us ({byts/tb/1e3:.0f} GB/s)", flush=True)
    if H * W <= 512 * 1024 and dt == torch.float32:
        ref = co.pers2equi(x.permute(0, 2, 3, 4, 1).cpu().numpy(), 80, nrows, P, (H, W))
        d = np.abs(a.cpu().numpy() - ref); d = d[np.isfinite(d)]
        print(f"     vs oracle: max {d.max():.2e}  frac>2e-4 {(d > 2e-4).mean():.1e}")
# variants at the benchmark shape
x = torch.rand((8, 18, 1, 256, 256), device=dev)
for pl in (8, 4, 2):
    for nb in (4, 2, 1):
        if nb > pl: continue
        L.set_option("p2e_planes", pl); L.set_option("p2e_nbuf", nb)
        t = timeit(lambda: pers2equi(x, 80, 4, 256, (512, 1024), None, layout=L.LAYOUT_BNCHW))
        print(f"  B=8 P=256 planes/wave {pl} slots {nb}: {t:.1f} us ({8*(256*256*18+512*1024)*4/t/1e3:.0f} GB/s)")
L.set_option("p2e_planes", 0); L.set_option("p2e_nbuf", 0)
# confidence blend
for (B, P) in [(8, 128), (2, 128), (8, 256)]:
    pr = torch.rand((B, 18, 1, P, P), device=dev) * 8; cf = torch.rand((B, 18, 1, P, P), device=dev)
    L.set_option("p2e_gather", 0)
    a = pers2equi_conf(pr * cf, cf, 80, 4, P, (512, 1024), layout=L.LAYOUT_BNCHW)
    ta = timeit(lambda: pers2equi_conf(pr, cf, 80, 4, P, (512, 1024), layout=L.LAYOUT_BNCHW))
    L.set_option("p2e_gather", 1)
    b = pers2equi_conf(pr * cf, cf, 80, 4, P, (512, 1024), layout=L.LAYOUT_BNCHW)
    tb = timeit(lambda: pers2equi_conf(pr, cf, 80, 4, P, (512, 1024), layout=L.LAYOUT_BNCHW))
    L.set_option("p2e_gather", 0)
    byts = B * (2 * P * P * 18 + 512 * 1024) * 4
    print(f"conf B={B} P={P}: equal={torch.equal(a, b)} lds {ta:.1f} us ({byts/ta/1e3:.0f} GB/s) gather {tb:.1f} us ({byts/tb/1e3:.0f} GB/s)")
