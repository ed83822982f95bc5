"""Throughput of plain calls (two half-batch lanes) vs pipelined(depth) at B panoramas per forward; bit check.
Environment: B (default 8), DEPTHS (default 1,2,3,4), GRAPHS=1 (one hipGraph per slot)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.model import spherical_model as sm
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
B = int(os.environ.get("B", "8"))
batches = [torch.rand((B, 3, 512, 1024), device="cuda") for _ in range(4)]
ref = [net(b, confidence=True).clone() for b in batches]
def plain(n):
    for i in range(n): net(batches[i % 4], confidence=True)
def piped(run, n):
    pend = []
    for i in range(n):
        pend.append(run(batches[i % 4], confidence=True))
        if len(pend) > run.depth: pend.pop(0).get()
    for p in pend: p.get()
plain(10); torch.cuda.synchronize(); t0 = time.perf_counter(); plain(40); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
print("B=%d plain            : %.3f ms/forward %.0f pano/s" % (B, dt * 1e3, B / dt), flush=True)
for depth in [int(x) for x in os.environ.get("DEPTHS", "1,2,3,4").split(",")]:
    run = net.pipelined(depth, graphs=os.environ.get('GRAPHS', '0') == '1')
    outs = [run(b, confidence=True) for b in batches]
    same = all(torch.equal(o.get(), r) for o, r in zip(outs, ref))
    piped(run, 10); torch.cuda.synchronize(); t0 = time.perf_counter(); piped(run, 60); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
    print("B=%d pipelined depth %d: %.3f ms/forward %.0f pano/s  bits %s" % (B, depth, dt * 1e3, B / dt, "same" if same else "DIFFER"), flush=True)
