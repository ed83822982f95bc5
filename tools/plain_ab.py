"""A/B of conv tile policies for plain calls (two half-batch lanes) and a lone panorama, interleaved in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
for B in (8, 1, 4):
    rgb = torch.rand((B, 3, 512, 1024), device="cuda")
    for _ in range(20): net(rgb)
    tiles = (-1, 5, 6, 7); acc = {t: [] for t in tiles}
    for rnd in range(5):
        for t in tiles:
            L.set_option("conv_sh_tile", t)
            for _ in range(5): net(rgb)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): net(rgb)
            torch.cuda.synchronize(); acc[t].append((time.perf_counter() - t0) / 40 * 1e3)
    for t in tiles:
        print("B=%d plain conv_sh_tile %2d: median %.3f ms/forward" % (B, t, sorted(acc[t])[2]), flush=True)
