"""Per-layer f16x3 conv microbench at the model's shapes (B=8 -> M=144).  ONLY=<idx,idx> selects rows."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
from omnifusion_amd.model._engine import split_weights_f16x3
lib = _lib.load()
if os.environ.get('LIBPATH'):                     # a variant of the library (tools/convabl.sh)
    lib = ctypes.CDLL(os.environ['LIBPATH']); lib.omni_last_error.restype = ctypes.c_char_p
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("M", "144"))
# (name, H, W, C1, C2, Cout, k, stride, pad, res)
CFGS = [("layer1", 32, 32, 64, 0, 64, 3, 1, 1, True), ("layer2_0a", 32, 32, 64, 0, 128, 3, 2, 1, False),
        ("layer2", 16, 16, 128, 0, 128, 3, 1, 1, True), ("layer3_0a", 16, 16, 128, 0, 256, 3, 2, 1, False),
        ("layer3", 8, 8, 256, 0, 256, 3, 1, 1, True), ("layer4_0a", 8, 8, 256, 0, 512, 3, 2, 1, False),
        ("layer4", 4, 4, 512, 0, 512, 3, 1, 1, True),
        ("de0_0", 8, 8, 512, 0, 256, 3, 1, 1, False), ("de0_1", 8, 8, 256, 256, 128, 3, 1, 1, False),
        ("de1_0", 16, 16, 128, 0, 128, 3, 1, 1, False), ("de1_1", 16, 16, 128, 128, 64, 3, 1, 1, False),
        ("de2_0", 32, 32, 64, 0, 64, 3, 1, 1, False), ("de2_1", 32, 32, 64, 64, 64, 3, 1, 1, False),
        ("de3_0", 64, 64, 64, 0, 64, 3, 1, 1, False), ("de3_1", 64, 64, 64, 64, 32, 3, 1, 1, False),
        ("de4_0", 128, 128, 32, 0, 32, 3, 1, 1, False)]
if os.environ.get("ONLY"):
    CFGS = [CFGS[int(i)] for i in os.environ["ONLY"].split(",")]
ws = torch.empty(64 << 20, device="cuda")
for name, H, W, C1, C2, Cout, k, s, pad, use_res in CFGS:
    x1 = torch.randn(M, H, W, C1, device="cuda")
    x2 = torch.randn(M, H, W, C2, device="cuda") if C2 else None
    K = (C1 + C2) * k * k
    w = torch.randn(Cout, K) / np.sqrt(K)
    w16 = split_weights_f16x3(w).cuda()
    b = torch.randn(Cout, device="cuda")
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    res = torch.randn(M, Ho, Wo, Cout, device="cuda") if use_res else None
    out = torch.empty(M, Ho, Wo, Cout, device="cuda")
    sk = int(os.environ.get("SPLITK", "0")) or lib.omni_conv2d_splitk_plan(M * Ho * Wo, Cout, K // 32)
    SH = os.environ.get("SH", "1") == "1"
    if SH:
        for tname in ("x1", "x2", "res"):
            tt_ = locals()[tname]
            if tt_ is not None:
                o_ = torch.empty_like(tt_); lib.omni_sh_from_f32(P(tt_), P(o_), ctypes.c_size_t(tt_.numel()), S())
                if tname == "x1": x1 = o_
                elif tname == "x2": x2 = o_
                else: res = o_
    fused_up = SH and os.environ.get("FUSED_UP", "1") == "1" and name in ("de2_0", "de3_0", "de4_0")    # as the model runs them: up-sampling inside the kernel
    if fused_up:
        xl = torch.randn(M, H // 2, W // 2, C1, device="cuda"); xls = torch.empty_like(xl)
        lib.omni_sh_from_f32(P(xl), P(xls), ctypes.c_size_t(xl.numel()), S())
    def run():
        if fused_up:
            rc = lib.omni_conv3x3_up2_sh_f16x3(P(xls), P(w16), P(b), P(out), int(os.environ.get("UPFMT", "1")), M, H // 2, W // 2, C1, Cout, 1, S())
            assert rc == 0, lib.omni_last_error()
            return
        if SH:
            rc = lib.omni_conv2d_sh_f16x3_ws(P(x1), P(x2), P(w16), P(b), P(res), P(out), 1, M, H, W, C1, C2, Cout, k, k, s, pad, 1,
                                             sk, P(ws), ctypes.c_size_t(ws.numel() * 4), S())
            assert rc == 0, lib.omni_last_error()
            return
        rc = lib.omni_conv2d_nhwc_f16x3_ws(P(x1), P(x2), P(w16), P(b), P(res), P(out), M, H, W, C1, C2, Cout, k, k, s, pad, 1,
                                           sk, P(ws), ctypes.c_size_t(ws.numel() * 4), S())
        assert rc == 0, lib.omni_last_error()
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    tt = e0.elapsed_time(e1) / 20 * 1e-3
    fl = 2 * M * Ho * Wo * Cout * K
    print("%-10s splitk=%d %7.1f us  %6.1f TF/s(fp32-equiv)  %4.1f%% of f16 peak" % (name, sk, tt * 1e6, fl / tt / 1e12, 3 * fl / tt / 2.5e15 * 100))
