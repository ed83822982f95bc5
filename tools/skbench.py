"""Split-K convolution launches at the model's shapes: the two-launch form (partials + sh_splitk_reduce_kernel) against the in-launch reduction
(omni_conv2d_sh_f16x3_sk_ws), per split factor.  M=<patches> (144 = 8 panoramas), ONLY=<names>, SPLITS=1,2,3,..., TILE=<conv_sh_tile>."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
from omnifusion_amd.model._engine import split_weights_f16x3
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("M", "144"))
# name, M scale, H, W, C1, C2, Cout, k, stride, res, fmt
CFGS = [("layer3", 1, 8, 8, 256, 0, 256, 3, 1, True, 1), ("layer3_0a", 1, 16, 16, 128, 0, 256, 3, 2, False, 1),
        ("layer4", 1, 4, 4, 512, 0, 512, 3, 1, True, 1), ("layer4_0a", 1, 8, 8, 256, 0, 512, 3, 2, False, 1),
        ("de0_0", 1, 8, 8, 512, 0, 256, 3, 1, False, 1), ("de0_1", 1, 8, 8, 256, 256, 128, 3, 1, False, 1),
        ("fc2", 1, 1, 1, 2048, 0, 512, 1, 1, True, 2), ("fc1", 1, 1, 1, 512, 0, 2048, 1, 1, False, 1), ("qkv", 1, 1, 1, 512, 0, 1536, 1, 1, False, 0),
        ("down", 1, 4, 4, 512, 0, 32, 1, 1, False, 0)]
if os.environ.get("ONLY"):
    CFGS = [c for c in CFGS if c[0] in os.environ["ONLY"].split(",")]
splits = [int(v) for v in os.environ.get("SPLITS", "1,2,3,4,5,7,9").split(",")]
if os.environ.get("TILE"):
    _lib.set_option("conv_sh_tile", int(os.environ["TILE"]))
tickets = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")


def sh(t):
    o = torch.empty_like(t)
    lib.omni_sh_from_f32(P(t), P(o), ctypes.c_size_t(t.numel()), ST())
    return o


def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(4e6))
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for name, _, H, W, C1, C2, Cout, k, s, use_res, fmt in CFGS:
    pad = k // 2
    x1 = sh(torch.randn(M, H, W, C1, device="cuda"))
    x2 = sh(torch.randn(M, H, W, C2, device="cuda")) if C2 else None
    K = (C1 + C2) * k * k
    w16 = split_weights_f16x3(torch.randn(Cout, K) / np.sqrt(K)).cuda()
    b = torch.randn(Cout, device="cuda")
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    rows = M * Ho * Wo
    res = torch.randn(M, Ho, Wo, Cout, device="cuda") if use_res else None
    if res is not None and not (fmt & 2):
        res = sh(res)
    out = torch.empty(M, Ho, Wo, Cout, device="cuda")
    plan2 = lib.omni_conv2d_splitk_plan(ctypes.c_longlong(rows), Cout, K // 32)
    plan1 = lib.omni_conv2d_sk_plan(ctypes.c_longlong(rows), Cout, K // 32, k, k, s, pad, H, W)
    line = "%-10s rows %6d ksteps %3d  two-launch plan %d, in-launch plan %d |" % (name, rows, K // 32, plan2, plan1)
    for S in splits:
        if S > max(1, K // 32 // 4) and S > 1:
            continue
        nb = max(int(lib.omni_conv2d_sk_ws_bytes(ctypes.c_longlong(rows), Cout, S)), 4)
        ws = torch.empty(nb // 4, device="cuda")

        def run(tk):
            rc = lib.omni_conv2d_sh_f16x3_sk_ws(P(x1), P(x2), P(w16), P(b), P(res), P(out), fmt, M, H, W, C1, C2, Cout, k, k, s, pad, 1,
                                                S, P(ws), ctypes.c_size_t(nb), P(tickets) if tk else None, ctypes.c_size_t(tickets.numel() if tk else 0), ST())
            assert rc == 0, lib.omni_last_error()
        t2 = timeit(lambda: run(False))
        t1 = timeit(lambda: run(True)) if S > 1 else t2
        line += "  S=%d: %5.1f / %5.1f" % (S, t2, t1)
    print(line + "   (us: two launches / in-launch)")
