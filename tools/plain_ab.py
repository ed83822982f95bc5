"""A/B of a library option (OPT, default conv_sh_tile; VALS) for plain calls (two half-batch lanes) and a lone panorama (BS, default 8,1,4), interleaved in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
OPT = os.environ.get("OPT", "conv_sh_tile")
for B in tuple(int(x) for x in os.environ.get("BS", "8,1,4").split(",")):
    rgb = torch.rand((B, 3, 512, 1024), device="cuda")
    for _ in range(20): net(rgb)
    tiles = tuple(int(x) for x in os.environ.get("VALS", "-1,5,6,7").split(",")); acc = {t: [] for t in tiles}
    for rnd in range(5):
        for t in tiles:
            L.set_option(OPT, t)
            for _ in range(5): net(rgb)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): net(rgb)
            torch.cuda.synchronize(); acc[t].append((time.perf_counter() - t0) / 40 * 1e3)
    for t in tiles:
        print("B=%d plain %s %2d: median %.3f ms/forward" % (B, OPT, t, sorted(acc[t])[2]), flush=True)
