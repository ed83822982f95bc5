"""Latency of ONE panorama's forward: eager back-to-back, hipGraph replay with a sync per request (true request latency), and the effect of
Engine.rows_gemm / Engine.latency_plan, interleaved in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.model._engine import Engine
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
one = torch.rand((1, 3, 512, 1024), device="cuda")
def measure(tag):
    for _ in range(10): net(one, confidence=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): net(one, confidence=True)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 100 * 1e3
    run = net.graphed(one, confidence=True)
    for _ in range(10): run(one)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): run(one); torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / 100 * 1e3
    t0 = time.perf_counter()
    for _ in range(100): run(one)
    torch.cuda.synchronize(); thr = (time.perf_counter() - t0) / 100 * 1e3
    return eager, lat, thr
res = {}
CASES = [("rows_gemm on", True, True), ("rows_gemm off", False, True), ("latency_plan off", False, False)]
for rnd in range(3):
    for tag, rg, lp in CASES:
        Engine.rows_gemm, Engine.latency_plan = rg, lp
        res.setdefault(tag, []).append(measure(tag))
for tag, _, _ in CASES:
    v = res[tag]
    med = lambda k: sorted(x[k] for x in v)[len(v) // 2]
    print("%-17s eager %.3f ms  graph, one request at a time %.3f ms  graph, back to back %.3f ms" % (tag, med(0), med(1), med(2)), flush=True)
