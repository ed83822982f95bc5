"""N forwards of the single-pass model at a given per-GPU batch and nothing else (for kernel-trace profiles: the trace's last
e2p..p2e kernel sequence IS a forward at this batch — bench.py also runs batch-1, host-fed and resample-pair legs)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--labels", default=None, help="write the label of every library call of the LAST forward (one per line) — one lane, so that "
                "the kernel order of the trace is the call order: tools/layerprof.py names the trace's rows with it")
ap.add_argument("--pipelined", type=int, default=0, help="D > 0: the forwards run as bench.py's timed region runs them — net.pipelined(D): whole-batch kernels, D forwards in flight")
a = ap.parse_args()
if a.labels:
    spherical_fusion.LANES = 1
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
rgb = torch.rand((a.batch, 3, 512, 1024), device="cuda")
from omnifusion_amd import _lib
if a.pipelined > 0:
    import collections
    run, pend = net.pipelined(a.pipelined), collections.deque()
    for k in range(a.steps):
        pend.append(run(rgb, confidence=True))
        if len(pend) > a.pipelined:
            pend.popleft().get()
    while pend:
        pend.popleft().get()
for k in range(0 if a.pipelined > 0 else a.steps):
    if a.labels and k == a.steps - 1:
        _lib.CALL_LOG = []
    net(rgb, confidence=True)
if a.labels:
    open(a.labels, "w").write("\n".join(_lib.CALL_LOG) + "\n")
torch.cuda.synchronize()
print("done", a.batch)
