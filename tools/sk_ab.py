"""A/B of Engine.in_launch_reduce (0 two launches | 1 in-launch reduction, same plan and bits | 2 in-launch with the unit plan) and of the split-K cap,
plain calls at BS panoramas, interleaved in one process; outputs of modes 0 and 1 compared bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.model._engine import Engine
from omnifusion_amd.weights import make_state_dict
L.set_option("conv_sk", 1)
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
for B in tuple(int(x) for x in os.environ.get("BS", "8,1,4").split(",")):
    rgb = torch.rand((B, 3, 512, 1024), device="cuda")
    cases = [("two launches", 0, 0), ("in-launch, same plan", 1, 0), ("in-launch, unit plan", 2, 0)]
    if B == 1:
        cases += [("two launches, cap %d" % c, 0, c) for c in (3, 4, 5, 6, 8)] + [("in-launch, cap %d" % c, 1, c) for c in (4, 5, 6, 8)]
    acc, outs = {c[0]: [] for c in cases}, {}
    for rnd in range(5):
        for tag, mode, cap in cases:
            Engine.in_launch_reduce = mode
            L.set_option("splitk_max", cap)
            net._lanes = None
            for _ in range(5): o = net(rgb)
            outs[tag] = o.clone()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): net(rgb)
            torch.cuda.synchronize(); acc[tag].append((time.perf_counter() - t0) / 40 * 1e3)
    assert torch.equal(outs["two launches"], outs["in-launch, same plan"])
    for tag, _, _ in cases:
        print("B=%d %-24s median %.3f ms/forward   max |d| vs two launches %.2e" % (B, tag, sorted(acc[tag])[2], (outs[tag] - outs["two launches"]).abs().max().item()), flush=True)
try:
    net._eng.read_overflow_flag()
    print("XCD premise held")
except RuntimeError as e:
    print("XCD premise VIOLATED (two lanes = two streams in flight):", str(e)[:90])
