"""A/B of a tuning option (OPT=name, VALS=v1,v2,..; default conv_sh_tile) in the pipelined mode, interleaved in ONE process
(boxes and thermal states differ by several %).  DEPTH (default 3), B (default 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
B = int(os.environ.get("B", "8")); depth = int(os.environ.get("DEPTH", "3"))
OPT = os.environ.get("OPT", "conv_sh_tile")
batches = [torch.rand((B, 3, 512, 1024), device="cuda") for _ in range(4)]
run = net.pipelined(depth)
def piped(n):
    pend = []
    for i in range(n):
        pend.append(run(batches[i % 4], confidence=True))
        if len(pend) > depth: pend.pop(0).get()
    for p in pend: p.get()
piped(40); torch.cuda.synchronize()
tiles = [int(x) for x in os.environ.get("VALS", "-1,0,3,4,5,7").split(",")]
acc = {t: [] for t in tiles}
for rnd in range(5):
    for t in tiles:
        L.set_option(OPT, t)
        piped(6); torch.cuda.synchronize(); t0 = time.perf_counter(); piped(40); torch.cuda.synchronize()
        acc[t].append(B * 40 / (time.perf_counter() - t0))
for t in tiles:
    v = acc[t]
    print("B=%d depth %d %s %2d: %s  median %.0f pano/s" % (B, depth, OPT, t, " ".join("%.0f" % x for x in v), sorted(v)[len(v) // 2]), flush=True)
