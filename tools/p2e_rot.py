"""pers2equi at the metric's shape (B = 8, 18 x 256^2 -> 512 x 1024) from HBM (rotating buffer sets, bench.py's `roofline_resample.frac`) against
cache-warm (one buffer set), per kernel option: tools/p2e_rot.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import kernel_us, kernel_us_rotating
from omnifusion_amd import _lib as L
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
lib = L.load(); dev = torch.device("cuda:0")
B, N, P, H, W = 8, 18, 256, 512, 1024
NSETS = 16 if "--walk" in sys.argv else 6      # (ablations: with a part switched off the other part's 6 sets fit the 256-MB memory-side cache)
sets = [torch.rand((B, N, 1, P, P), device=dev) for _ in range(NSETS)]
f = lambda d: pers2equi(d, (80, 80), 4, (P, P), (H, W), None, layout=L.LAYOUT_BNCHW)
DEFAULTS = {"p2e_walk": 1, "p2e_nbuf": 0, "p2e_planes": 0, "p2e_lds_kb": 0}
VARIANTS = [{}, {"p2e_nbuf": 4}, {"p2e_walk": 2}, {"p2e_planes": 4}, {"p2e_planes": 4, "p2e_nbuf": 4}, {"p2e_walk": 2, "p2e_planes": 4}, {"p2e_nbuf": 1}]
if "--occ" in sys.argv: VARIANTS = [{}, {"p2e_lds_kb": 12}, {"p2e_lds_kb": 13}, {"p2e_lds_kb": 16}, {"p2e_lds_kb": 20}, {"p2e_lds_kb": 26}, {"p2e_lds_kb": 40}, {"p2e_lds_kb": 20, "p2e_nbuf": 4}]   # 16 | 13 | 12 | 10 | 8 | 6 | 4 waves per CU
if "--walk" in sys.argv: VARIANTS = [{"p2e_walk": 2}]          # (debug build + OMNI_P2E_DBG bits: tools/p2e_rot_ablate.sh)
for rnd in range(2):
    for v in VARIANTS:
        for k, d in DEFAULTS.items(): L.set_option(k, v.get(k, d))
        lib.omni_geometry_cache_clear()
        warm = kernel_us([lambda: f(sets[0])], dev, 20)[0]
        rot = kernel_us_rotating(f, sets, dev, 20)
        print(f"round {rnd} {str(v):44s} cache-warm {warm * 1e6:6.2f} us   rotating {rot * 1e6:6.2f} us", flush=True)
for k, d in DEFAULTS.items(): L.set_option(k, d)
