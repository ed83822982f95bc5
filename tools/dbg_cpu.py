import sys, time, torch
sys.path.insert(0, '.')
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
for B in (1, 8):
    rgb = torch.rand(B, 3, 512, 1024, device="cuda")
    for _ in range(3): net(rgb)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); net(rgb); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: CPU enqueue of one forward {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms")
    run = net.graphed(rgb)
    for _ in range(3): run(rgb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): run(rgb)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: graphed {1e3*(t2-t0)/20:.3f} ms/step ({B*20/(t2-t0):.0f}/s)")
    t0 = time.perf_counter()
    for _ in range(20): net(rgb)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: eager   {1e3*(t2-t0)/20:.3f} ms/step ({B*20/(t2-t0):.0f}/s)")
