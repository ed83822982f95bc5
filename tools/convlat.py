"""Latency of ONE deep-layer convolution at batch 1 (M = 18 patches) with COLD weights: the launches rotate over NCOPY weight sets so that
no launch finds its weights in L2 (as in a real forward, where every layer's weights are touched once); replayed from a hipGraph so that the
host's launch rate does not hide the device time.  Rows: 3 / 6 LDS stages of the tile kernel; columns: split-K factor; for the
transformer's GEMMs the register-streaming rows kernel beside them."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
from omnifusion_amd.model._engine import split_weights_f16x3
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("M", "18")); NCOPY = int(os.environ.get("NCOPY", "16"))
CFGS = [("layer3", 8, 8, 256, 256, 3, 1, 1), ("layer4", 4, 4, 512, 512, 3, 1, 1), ("proj", 1, 1, 512, 512, 1, 1, 0), ("fc1", 1, 1, 512, 2048, 1, 1, 0), ("fc2", 1, 1, 2048, 512, 1, 1, 0),
        ("qkv", 1, 1, 512, 1536, 1, 1, 0)]
ws = torch.empty(64 << 20, device="cuda")
for name, H, W, C1, Cout, k, s, pad in CFGS:
    x = torch.randn(M, H, W, C1, device="cuda"); xs = torch.empty_like(x)
    lib.omni_sh_from_f32(P(x), P(xs), ctypes.c_size_t(x.numel()), S())
    K = C1 * k * k
    w16 = [split_weights_f16x3(torch.randn(Cout, K) / np.sqrt(K)).cuda() for _ in range(NCOPY)]
    b = torch.randn(Cout, device="cuda")
    out = torch.empty(M, H, W, Cout, device="cuda")
    for nodeep in (1, 0):
        lib.omni_set_option(b"conv_nodeep", nodeep)
        line = "%-7s stages %s:" % (name, {1: "3", 0: "6"}[nodeep])
        for sk in (1, 2, 3, 4, 6, 8, 12, 16):
            if sk > K // 32 // 2: continue
            def run(i):
                rc = lib.omni_conv2d_sh_f16x3_ws(P(xs), None, P(w16[i % NCOPY]), P(b), None, P(out), 1, M, H, W, C1, 0, Cout, k, k, s, pad, 1,
                                                 sk, P(ws), ctypes.c_size_t(ws.numel() * 4), S())
                assert rc == 0, lib.omni_last_error()
            for i in range(NCOPY): run(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                run(0); torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=st):
                    for i in range(4 * NCOPY): run(i)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): g.replay()
            e1.record(); torch.cuda.synchronize()
            line += "  S%d %.1f" % (sk, e0.elapsed_time(e1) / (12 * NCOPY) * 1e3)
        print(line + "  us (conv + reduce)", flush=True)
    if k == 1 and K in (512, 2048):
        o2 = torch.empty(M, Cout, device="cuda")
        w16 = [w.clone() for w in w16]          # (timing only: the unpacked bytes stand in for packed ones)
        def run(i):
            assert lib.omni_gemm_rows_sh_f16x3(P(xs), P(w16[i % NCOPY]), P(b), None, P(o2), 0, M, K, Cout, 1, S()) == 0, lib.omni_last_error()
        for i in range(NCOPY): run(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            run(0); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=st):
                for i in range(4 * NCOPY): run(i)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): g.replay()
        e1.record(); torch.cuda.synchronize()
        print("%-7s rows kernel: %.1f us" % (name, e0.elapsed_time(e1) / (12 * NCOPY) * 1e3), flush=True)
