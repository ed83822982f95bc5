"""Batch-1..3 latency of the single-pass forward vs the batch the split-K factors are planned for."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.model import _engine
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
for B in (1, 2, 3):
    rgb = torch.rand((B, 3, 512, 1024), generator=torch.Generator().manual_seed(B)).cuda()
    ref = None
    for nominal in (8, 6, 5, 4, 3, 2, 1):
        _engine.Engine.NOMINAL_BATCH = nominal
        for _ in range(10): o = net(rgb, confidence=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(60): o = net(rgb, confidence=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
        if ref is None: ref = o.clone()
        print("B=%d split-K planned for batch %d: %.3f ms per forward   max|diff| vs plan 8: %.2g" % (B, nominal, dt * 1e3, (o - ref).abs().max().item()), flush=True)
