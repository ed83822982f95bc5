"""Is the pipelined mode host-bound?  Per-slot hipGraph replay vs eager submission."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion, _concurrent_streams
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
B = 8
batches = [torch.rand((B, 3, 512, 1024), device="cuda") for _ in range(4)]
ref = [net(b, confidence=True).clone() for b in batches]
for depth in (2, 3):
    run = net.pipelined(depth)
    for b in batches: run(b, confidence=True).get()
    torch.cuda.synchronize()
    # eager
    def eager(n):
        pend = []
        for i in range(n):
            pend.append(run(batches[i % 4], confidence=True))
            if len(pend) > depth: pend.pop(0).get()
        for p in pend: p.get()
    eager(10); torch.cuda.synchronize(); t0 = time.perf_counter(); eager(60); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
    print("depth %d eager : %.3f ms/forward %.0f pano/s" % (depth, dt * 1e3, B / dt), flush=True)
    # graphs: one per slot
    slots = []
    spherical_fusion.LANES = 1
    main = net._eng
    for eng, st in run.slots:
        static_in = batches[0].clone()
        net._eng = eng
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            net(static_in, confidence=True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            static_out = net(static_in, confidence=True)
        slots.append((g, st, static_in, static_out))
    net._eng = main
    spherical_fusion.LANES = 2
    torch.cuda.synchronize()
    outs = []
    def graphed(n, check=False):
        for i in range(n):
            g, st, sin, sout = slots[i % depth]
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                sin.copy_(batches[i % 4], non_blocking=True)
                g.replay()
                if check: outs.append((i % 4, sout.clone()))
    graphed(8, check=True); torch.cuda.synchronize()
    print("   graph outputs equal:", all(torch.equal(o, ref[k]) for k, o in outs))
    graphed(10); torch.cuda.synchronize(); t0 = time.perf_counter(); graphed(60); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
    print("depth %d graphs: %.3f ms/forward %.0f pano/s" % (depth, dt * 1e3, B / dt), flush=True)
