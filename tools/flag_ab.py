"""A/B of a boolean/int Engine class attribute (FLAG=name, VALS=a,b,..) or of a library option (FLAG=opt:name), interleaved in one process: pipelined (depth 3) and plain calls at
B panoramas, plus the lone-panorama latency."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.model._engine import Engine
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
B = int(os.environ.get("B", "8")); depth = 3
FLAG = os.environ.get("FLAG", "fuse_up"); VALS = [int(v) for v in os.environ.get("VALS", "1,0").split(",")]
batches = [torch.rand((B, 3, 512, 1024), device="cuda") for _ in range(4)]
one = batches[0][:1].contiguous()
run = net.pipelined(depth)
def piped(n):
    pend = []
    for i in range(n):
        pend.append(run(batches[i % 4], confidence=True))
        if len(pend) > depth: pend.pop(0).get()
    for p in pend: p.get()
def plain(n, x=None):
    for i in range(n): net(batches[i % 4] if x is None else x, confidence=True)
ref = None
acc = {v: ([], [], []) for v in VALS}
for rnd in range(4):
    for v in VALS:
        if FLAG.startswith("opt:"):                                 # a library option (OMNI_* switch) instead of an Engine attribute
            from omnifusion_amd import _lib; _lib.set_option(FLAG[4:], v)
        else: setattr(Engine, FLAG, type(getattr(Engine, FLAG))(v))
        o = net(batches[0], confidence=True)
        if ref is None: ref = o.clone()
        same = torch.equal(o, ref)
        piped(6); torch.cuda.synchronize(); t0 = time.perf_counter(); piped(30); torch.cuda.synchronize()
        acc[v][0].append(B * 30 / (time.perf_counter() - t0))
        plain(4); torch.cuda.synchronize(); t0 = time.perf_counter(); plain(20); torch.cuda.synchronize()
        acc[v][1].append(B * 20 / (time.perf_counter() - t0))
        plain(10, one); torch.cuda.synchronize(); t0 = time.perf_counter(); plain(60, one); torch.cuda.synchronize()
        acc[v][2].append((time.perf_counter() - t0) / 60 * 1e3)
        if rnd == 0: print("%s=%d: output %s the first value's" % (FLAG, v, "equals" if same else "DIFFERS from"), flush=True)
med = lambda x: sorted(x)[len(x) // 2]
for v in VALS:
    print("B=%d %s=%d: pipelined %.0f pano/s   plain %.0f pano/s   one panorama %.3f ms" % (B, FLAG, v, med(acc[v][0]), med(acc[v][1]), med(acc[v][2])), flush=True)
