"""VERDICT r2 #7, one measured experiment: can Winograd F(2x2, 3x3) on the f16x3 path beat the direct convolutions by >= 1.3x?

An UPPER BOUND for it, assembled from kernels that exist: the multiply stage of F(2x2,3x3) is 16 independent GEMMs
[tiles x Cin] x [Cin x Cout] (tiles = M * H/2 * W/2), timed here as ONE 1x1 convolution launch of conv_sh_kernel over 16 * tiles rows
with fp32 output (the inverse transform needs the 16 products unrounded) — the same matrix work and operand traffic as 16 weight sets
would have — plus the two transforms priced at their bare memory traffic by device copies of the same byte counts (input transform:
read the SH tensor, write 16 / 4 = 4x its size as SH; output transform: read 16 fp32 values per tile and channel, write the SH output).
Against: the direct 3x3 convolution as the model runs it (tile or halo kernel, fused epilogue).
"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
from omnifusion_amd.model._engine import split_weights_f16x3
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("M", "144"))
CFGS = [("layer1", 32, 32, 64, 64), ("layer2", 16, 16, 128, 128), ("layer3", 8, 8, 256, 256), ("layer4", 4, 4, 512, 512),
        ("de1_0", 16, 16, 128, 128), ("de2_1a", 32, 32, 128, 64), ("de3_1a", 64, 64, 128, 32)]
ws = torch.empty(64 << 20, device="cuda")


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); torch.cuda._sleep(int(8e6))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def sh(t):
    o = torch.empty_like(t); lib.omni_sh_from_f32(P(t), P(o), ctypes.c_size_t(t.numel()), S()); return o


print(f"# M = {M} patches (8 panoramas); times in us; 'bound' = multiply stage + transform traffic, an optimistic estimate of a Winograd layer")
for name, H, W, Cin, Cout in CFGS:
    x = sh(torch.randn(M, H, W, Cin, device="cuda"))
    w3 = split_weights_f16x3(torch.randn(Cout, Cin * 9) / np.sqrt(Cin * 9)).cuda()
    w1 = split_weights_f16x3(torch.randn(Cout, Cin) / np.sqrt(Cin)).cuda()
    b = torch.randn(Cout, device="cuda")
    out = torch.empty(M, H, W, Cout, device="cuda")
    sk = lib.omni_conv2d_splitk_plan(M * H * W, Cout, Cin * 9 // 32)
    def direct():
        rc = lib.omni_conv2d_sh_f16x3_ws(P(x), None, P(w3), P(b), None, P(out), 1, M, H, W, Cin, 0, Cout, 3, 3, 1, 1, 1, sk, P(ws), ctypes.c_size_t(ws.numel() * 4), S())
        assert rc == 0, lib.omni_last_error()
    tiles = M * (H // 2) * (W // 2)
    rows = 16 * tiles
    v = sh(torch.randn(rows, 1, 1, Cin, device="cuda"))
    mo = torch.empty(rows, 1, 1, Cout, device="cuda")
    def mult():
        rc = lib.omni_conv2d_sh_f16x3_ws(P(v), None, P(w1), None, None, P(mo), 0, rows, 1, 1, Cin, 0, Cout, 1, 1, 1, 0, 0, 1, None, ctypes.c_size_t(0), S())
        assert rc == 0, lib.omni_last_error()
    t_dir, t_mul = timeit(direct), timeit(mult)
    # transform traffic: in = read x + write v (both SH: 4 B per element); out = read mo (fp32) + write out (SH)
    a1 = torch.empty((x.numel() + v.numel()) // 2, device="cuda"); a2 = torch.empty_like(a1)
    c1 = torch.empty((mo.numel() + out.numel()) // 2, device="cuda"); c2 = torch.empty_like(c1)
    t_in, t_out = timeit(lambda: a2.copy_(a1)), timeit(lambda: c2.copy_(c1))
    fl = 2 * M * H * W * Cout * Cin * 9
    print(f"{name:8s} {H:3d}x{W:<3d} {Cin:3d}->{Cout:<3d}: direct {t_dir:7.1f} ({fl / t_dir / 1e6:5.0f} TF/s) | multiply stage {t_mul:7.1f} + transforms {t_in:6.1f} + {t_out:6.1f} = bound {t_mul + t_in + t_out:7.1f} "
          f"-> at best {t_dir / (t_mul + t_in + t_out):4.2f}x (multiply stage alone {t_dir / t_mul:4.2f}x)")
