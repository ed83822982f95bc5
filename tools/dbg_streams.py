import sys, time, torch
sys.path.insert(0, '.')
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sd = make_state_dict(42, 18, False)
nets = []
for i in range(NS):
    n = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); n.load_state_dict(sd); nets.append(n)
rgb = torch.rand(B, 3, 512, 1024, device="cuda")
streams = [torch.cuda.Stream() for _ in range(NS)]
chunks = rgb.chunk(NS)
def fwd_multi():
    for n, s, c in zip(nets, streams, chunks):
        with torch.cuda.stream(s):
            n(c, confidence=True)
def fwd_single():
    nets[0](rgb, confidence=True)
for name, f in (("single", fwd_single), ("multi", fwd_multi)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"B={B} {name} NS={NS}: {dt*1e3:.3f} ms/step  {B/dt:.0f} pano/s")
# graphed variants
g = [n.graphed(c, confidence=True) for n, c in zip(nets, chunks)]
def fwd_multi_graph():
    for run, s, c in zip(g, streams, chunks):
        with torch.cuda.stream(s):
            run(c)
try:
    for _ in range(3): fwd_multi_graph()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): fwd_multi_graph()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"B={B} multi-graph NS={NS}: {dt*1e3:.3f} ms/step  {B/dt:.0f} pano/s")
except Exception as e:
    print("graph variant failed:", e)
