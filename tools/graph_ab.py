import os, sys, time, collections
sys.path.insert(0, os.getcwd())
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
rgb = torch.rand((8, 3, 512, 1024), device="cuda")
for _ in range(10): net(rgb)
# host time to enqueue one forward
torch.cuda.synchronize(); run = net.pipelined(3)
for mode in ("eager", "graphs"):
    run = net.pipelined(3, graphs=(mode == "graphs"))
    pend = collections.deque()
    def step():
        pend.append(run(rgb, confidence=True))
        if len(pend) > 3: pend.popleft().get()
    for _ in range(30): step()
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100): step()
        while pend: pend.popleft().get()
        torch.cuda.synchronize()
        res.append(8 * 100 / (time.perf_counter() - t0))
    print(mode, "pipelined(3):", ["%.0f" % r for r in res], "panoramas/s", flush=True)
# pure host enqueue time (GPU far behind): enqueue 20 forwards without waiting
run = net.pipelined(3)
torch.cuda.synchronize(); t0 = time.perf_counter()
ps = [run(rgb, confidence=True) for _ in range(20)]
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host enqueue per forward: %.3f ms" % ((t1 - t0) / 20 * 1e3))
