"""Which network kernel disturbs pers2equi?  Victim: plain p2e (gather path) on stream 1; noise: ONE kind of kernel on stream 2."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model._engine import split_weights_f16x3
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
lib = L.load()
Pp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
B, N, P = 8, 18, 128
lay = L.LAYOUT_BNCHW
gather = int(os.environ.get("GATHER", "1"))
L.set_option("p2e_gather", gather)
a0 = torch.rand((B, N, 1, P, P), device="cuda")
victim = lambda: pers2equi(a0, 80, 4, P, (512, 1024), None, layout=lay)
ref = victim().clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
M = 144
def sh(t):
    o = torch.empty_like(t); lib.omni_sh_from_f32(Pp(t), Pp(o), ctypes.c_size_t(t.numel()), L.stream_of(t)); return o
def conv_noise(H, W, C1, Cout, k=3, stride=1, splitk=1, sh_in=True, pad=None):
    x = sh(torch.randn(M, H, W, C1, device="cuda")); K = k * k * C1
    w16 = split_weights_f16x3(torch.randn(Cout, K) / np.sqrt(K)).cuda(); b = torch.randn(Cout, device="cuda")
    pad = k // 2 if pad is None else pad
    Ho = (H + 2 * pad - k) // stride + 1
    out = torch.empty(M, Ho, Ho, Cout, device="cuda"); ws = torch.empty(max(1, splitk) * M * Ho * Ho * Cout, device="cuda")
    def run():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.omni_conv2d_sh_f16x3_ws(Pp(x), None, Pp(w16), Pp(b), None, Pp(out), 1, M, H, W, C1, 0, Cout, k, k, stride, pad, 1, splitk, Pp(ws), ctypes.c_size_t(ws.numel() * 4), st)
        assert rc == 0
    return run
x64 = sh(torch.randn(M, 32, 32, 64, device="cuda")); up_out = torch.empty(M, 64, 64, 64, device="cuda")
mp_in = sh(torch.randn(M, 64, 64, 64, device="cuda")); mp_out = torch.empty(M, 32, 32, 64, device="cuda")
patches = torch.rand((M, 3, 128, 128), device="cuda")
noises = {
    "none": lambda: None,
    "halo conv 64->64 @32": conv_noise(32, 32, 64, 64),
    "halo conv 32->32 @128": conv_noise(128, 128, 32, 32),
    "conv_sh 128->128 @16": conv_noise(16, 16, 128, 128),
    "conv_sh 128->128 @16 pad 0 (no out-of-range lanes)": conv_noise(16, 16, 128, 128, pad=0),
    "conv_sh 1x1 128->128 @16": conv_noise(16, 16, 128, 128, k=1),
    "e2p (LDS path, has out-of-range lanes)": None,
}
from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
rgb = torch.rand((B, 3, 512, 1024), device='cuda')
noises['e2p (LDS path, has out-of-range lanes)'] = lambda: equi2pers_patches(rgb, 80, 4, 256, layout=lay)
ONLY = os.environ.get("NOISE")
for name, nz in noises.items():
    if ONLY and not any(o in name for o in ONLY.split("|")): continue
    bad = 0; nb = 0
    for rep in range(10):
        with torch.cuda.stream(s2):
            for _ in range(40): nz()
        with torch.cuda.stream(s1):
            outs = [victim() for _ in range(12)]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref): bad += 1; nb += int((o != ref).sum())
    print("noise %-52s wrong %3d / 120  pixels %d" % (name, bad, nb), flush=True)
