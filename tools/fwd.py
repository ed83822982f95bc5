"""N forwards of the single-pass model at a given per-GPU batch and nothing else (for kernel-trace profiles: the trace's last
e2p..p2e kernel sequence IS a forward at this batch — bench.py also runs batch-1, host-fed and resample-pair legs)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
rgb = torch.rand((a.batch, 3, 512, 1024), device="cuda")
for _ in range(a.steps):
    net(rgb, confidence=True)
torch.cuda.synchronize()
print("done", a.batch)
