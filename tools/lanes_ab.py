"""Plain calls net(rgb) at 8 panoramas (test.py's loop as it is written) against the number of lanes the batch is split over inside a forward
(spherical_fusion.LANES; default 2): tools/lanes_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
B = int(os.environ.get("B", "8"))
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda().eval(); net.load_state_dict(make_state_dict(42, 18, False))
rgb = torch.rand((B, 3, 512, 1024), device="cuda")
ref = None
for rnd in range(2):
    for lanes in (2, 1, 3, 4, 2):
        net.LANES = lanes; net._lanes = None
        for _ in range(5): out = net(rgb)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(40): out = net(rgb)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 40
        d = out[0] if isinstance(out, (tuple, list)) else out
        if ref is None: ref = d.clone()
        print(f"round {rnd} lanes {lanes}: {dt * 1e3:.3f} ms per forward, {B / dt:6.0f} panoramas/s, same bits as the first: {torch.equal(d, ref)}", flush=True)
