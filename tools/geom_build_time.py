"""Cost of building a geometry (tables of both operators) with and without the 8-row pers2equi tile set (option p2e_tile8): tools/geom_build_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
lib = L.load()
for (H, W, nrows, P) in ((512, 1024, 4, 256), (1024, 2048, 6, 256), (2048, 4096, 6, 512)):
    N = {4: 18, 6: 46}[nrows]
    x = torch.rand((1, N, 1, P, P), device="cuda")
    for v in (1, 0, 1, 0):
        L.set_option("p2e_tile8", v); lib.omni_geometry_cache_clear(); torch.cuda.synchronize()
        t = time.perf_counter(); pers2equi(x, (80, 80), nrows, (P, P), (H, W), None, layout=L.LAYOUT_BNCHW); torch.cuda.synchronize()
        print(f"{H}x{W} nrows {nrows} P {P}: p2e_tile8={v} first call {1e3 * (time.perf_counter() - t):.1f} ms")
L.set_option("p2e_tile8", 1)
