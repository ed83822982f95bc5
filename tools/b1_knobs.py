"""One panorama per forward: ms per forward against the engine's latency knobs, interleaved in one process (tools/b1_knobs.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model._engine import Engine
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
one = torch.rand((1, 3, 512, 1024), device="cuda")
def t(n=60):
    for _ in range(10): net(one, confidence=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): net(one, confidence=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
variants = [("default", {}), ("SINGLE_BATCH=2", {"SINGLE_BATCH": 2}), ("SINGLE_BATCH=3", {"SINGLE_BATCH": 3}), ("SINGLE_BATCH=6", {"SINGLE_BATCH": 6}), ("SINGLE_BATCH=8", {"SINGLE_BATCH": 8}),
            ("rows_gemm=False", {"rows_gemm": False}), ("fuse_ln=False", {"fuse_ln": False})]
acc = {n: [] for n, _ in variants}
for rnd in range(4):
    for n, kv in variants:
        old = {k: getattr(Engine, k) for k in kv}
        for k, v in kv.items(): setattr(Engine, k, v)
        acc[n].append(t())
        for k, v in old.items(): setattr(Engine, k, v)
for n, _ in variants:
    print(f"{n:20s} {' '.join('%.3f' % x for x in acc[n])}  median {sorted(acc[n])[len(acc[n]) // 2]:.3f} ms", flush=True)
