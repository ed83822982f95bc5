"""How to time a 20-40 us kernel with HIP events: per-launch brackets (queued behind device spinning) vs one bracket around N launches.
Run under rocprofv3 --kernel-trace --stats to compare both with the dispatch time stamps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from omnifusion_amd import _lib as L
from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
dev = torch.device("cuda:0")
B, P = 8, 256
x = torch.rand((B, 3, 512, 1024), device=dev); d = torch.rand((B, 18, 1, P, P), device=dev)
f1 = lambda: equi2pers_patches(x, 80, 4, (P, P), layout=L.LAYOUT_BNCHW)
f2 = lambda: pers2equi(d, 80, 4, (P, P), (512, 1024), None, layout=L.LAYOUT_BNCHW)
for _ in range(200): f1(); f2()
torch.cuda.synchronize()
for rep in range(3):
    a = bench.kernel_us([f1, f2], dev, 20)
    a1 = bench.kernel_us([f1], dev, 20); a2 = bench.kernel_us([f2], dev, 20)
    def many(f, n=20):
        torch.cuda.synchronize(); torch.cuda._sleep(int(12e6))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    b1, b2 = many(f1), many(f2)
    print(f"per-launch brackets, alternating: e2p {a[0]*1e6:.1f} p2e {a[1]*1e6:.1f} | per-launch brackets, one kernel: {a1[0]*1e6:.1f} {a2[0]*1e6:.1f} | one bracket around 20 launches: {b1*1e6:.1f} {b2*1e6:.1f} us")
