"""The halo kernel's small-image mode (OMNI_CONV_IMG) against the im2col tile kernel: same operator, K summed in a different order."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
from omnifusion_amd.model._engine import split_weights_f16x3
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
def sh(t):
    o = torch.empty_like(t); lib.omni_sh_from_f32(P(t), P(o), ctypes.c_size_t(t.numel()), S()); return o
for (M, HW, C1, C2, Cout, use_res) in [(144, 16, 128, 0, 128, True), (144, 8, 256, 0, 256, True), (144, 4, 512, 0, 512, True), (72, 8, 256, 256, 128, False),
                                       (18, 16, 128, 128, 64, False), (8, 4, 512, 0, 512, True), (2, 8, 64, 0, 64, False)]:
    x1 = torch.randn(M, HW, HW, C1, device="cuda"); x2 = torch.randn(M, HW, HW, C2, device="cuda") if C2 else None
    K = (C1 + C2) * 9
    w = torch.randn(Cout, K) / np.sqrt(K); w16 = split_weights_f16x3(w).cuda(); b = torch.randn(Cout, device="cuda")
    res = torch.randn(M, HW, HW, Cout, device="cuda") if use_res else None
    outs = []
    for img in (0, 1):
        _lib.set_option("conv_img", img)
        out = torch.empty(M, HW, HW, Cout, device="cuda")
        s1, s2, sr = sh(x1), (sh(x2) if C2 else None), (sh(res) if use_res else None)
        rc = lib.omni_conv2d_sh_f16x3_ws(P(s1), P(s2), P(w16), P(b), P(sr), P(out), 0, M, HW, HW, C1, C2, Cout, 3, 3, 1, 1, 1,
                                         1, None, ctypes.c_size_t(0), S())
        assert rc == 0, lib.omni_last_error()
        torch.cuda.synchronize(); outs.append(out)
    # fp64 reference on the CPU for a few images
    n = min(M, 2)
    xx = torch.cat([x1[:n]] + ([x2[:n]] if C2 else []), dim=3).double().cpu().permute(0, 3, 1, 2)
    wk = w.double().view(Cout, 9, C1 + C2)                       # K order (tap, channel)
    wc = wk.view(Cout, 3, 3, C1 + C2).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xx, wc, b.double().cpu(), padding=1)
    if use_res: ref = ref + res[:n].double().cpu().permute(0, 3, 1, 2)
    ref = ref.clamp_min(0).permute(0, 2, 3, 1)
    d01 = (outs[0] - outs[1]).abs().max().item()
    e0 = (outs[0][:n].double().cpu() - ref).abs().max().item(); e1 = (outs[1][:n].double().cpu() - ref).abs().max().item()
    print(f"M={M} {HW}x{HW} C={C1}+{C2}->{Cout}: |tiles - halo| {d01:.2e}; vs fp64: tiles {e0:.2e} halo {e1:.2e}  {'OK' if e1 < 2e-5 and d01 < 2e-5 else 'BAD'}")
