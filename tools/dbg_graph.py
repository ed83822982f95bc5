import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
for B in (1, 8):
    rgb = torch.rand(B, 3, 512, 1024, device='cuda')
    ref = net(rgb).clone()
    run = net.graphed(rgb)
    out = run(rgb)
    print("B", B, "graph == eager:", torch.equal(out, ref))
    for name, fn in (("eager", lambda: net(rgb)), ("graph", lambda: run(rgb))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"   {name}: {dt*1e3:.3f} ms/forward  {B/dt:.1f} pano/s")
