"""Plain calls net(rgb) at 8 panoramas (two half-batch lanes on two streams) in a FRESH process: ms per forward.  Run several times — the lanes' streams are chosen by
measurement once per process (spherical_model._concurrent_streams); a process whose lanes share a hardware queue would show here as a slow outlier."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
rgb = torch.rand((8, 3, 512, 1024), device="cuda")
pre = int(os.environ.get("PRE_STREAMS", "0"))
keep = [torch.cuda.Stream() for _ in range(pre)]                    # streams created by "other code" before the model's first forward
for _ in range(20): net(rgb, confidence=True)
torch.cuda.synchronize(); ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(40): net(rgb, confidence=True)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40 * 1e3)
print(f"pre-created streams {pre}: {' '.join('%.3f' % t for t in ts)} ms per forward", flush=True)
