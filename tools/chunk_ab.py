"""A/B of Engine.tail_chunk / Engine.front_chunk (panoramas per pass through the widest stages), interleaved in one process:
pipelined (depth 3) and plain calls at B panoramas."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.model._engine import Engine
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
B = int(os.environ.get("B", "8")); depth = 3
batches = [torch.rand((B, 3, 512, 1024), device="cuda") for _ in range(4)]
run = net.pipelined(depth)
def piped(n):
    pend = []
    for i in range(n):
        pend.append(run(batches[i % 4], confidence=True))
        if len(pend) > depth: pend.pop(0).get()
    for p in pend: p.get()
def plain(n):
    for i in range(n): net(batches[i % 4], confidence=True)
CASES = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("CASES", "0:0,0:2,2:0,2:2,1:1,4:4").split(",")]
ref = None
acc = {c: ([], []) for c in CASES}
for rnd in range(4):
    for c in CASES:
        Engine.tail_chunk, Engine.front_chunk = c
        o = net(batches[0], confidence=True)
        if ref is None: ref = o.clone()
        assert torch.equal(o, ref), c
        piped(6); torch.cuda.synchronize(); t0 = time.perf_counter(); piped(30); torch.cuda.synchronize()
        acc[c][0].append(B * 30 / (time.perf_counter() - t0))
        plain(4); torch.cuda.synchronize(); t0 = time.perf_counter(); plain(20); torch.cuda.synchronize()
        acc[c][1].append(B * 20 / (time.perf_counter() - t0))
med = lambda v: sorted(v)[len(v) // 2]
for c in CASES:
    print("B=%d tail_chunk %d front_chunk %d: pipelined %.0f pano/s   plain %.0f pano/s" % (B, c[0], c[1], med(acc[c][0]), med(acc[c][1])), flush=True)
