"""Investigation: tools/pk/pk_probe.hip beside a halo convolution of the product library — which packed instruction forms go wrong, in which lanes.
build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -ffp-contract=off pk_probe.hip -o libpk_probe.so   (the assembler refuses the
inline packed instructions when the feature is switched off, so this file is built WITH it)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model._engine import split_weights_f16x3
lib = L.load()
pk = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpk_probe.so"))
pk.pk_probe.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p)]
Pp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
M = 144
def sh(t):
    o = torch.empty_like(t); lib.omni_sh_from_f32(Pp(t), Pp(o), ctypes.c_size_t(t.numel()), L.stream_of(t)); return o
x = sh(torch.randn(M, 128, 128, 32, device="cuda")); K = 9 * 32
w16 = split_weights_f16x3(torch.randn(32, K) / np.sqrt(K)).cuda(); b = torch.randn(32, device="cuda")
out = torch.empty(M, 128, 128, 32, device="cuda"); ws = torch.empty(4, device="cuda")
def conv():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.omni_conv2d_sh_f16x3_ws(Pp(x), None, Pp(w16), Pp(b), None, Pp(out), 1, M, 128, 128, 32, 0, 32, 3, 3, 1, 1, 1, 1, Pp(ws), ctypes.c_size_t(16), st) == 0
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = torch.zeros(64, dtype=torch.int32, device="cuda")
BLOCKS, ITERS = 4096, 400
which = 0
print("checks per run: %d blocks x 256 lanes x %d trips x 2 halves, 6 x 8 launches" % (BLOCKS, ITERS))
while True:
    name = ctypes.c_char_p()
    if pk.pk_probe(which, None, 0, 0, None, ctypes.byref(name)) != 0: break
    res = []
    for noise in (False, True):
        bad.zero_(); torch.cuda.synchronize()
        for rep in range(6):
            if noise:
                with torch.cuda.stream(s2):
                    for _ in range(40): conv()
            with torch.cuda.stream(s1):
                for _ in range(8):
                    assert pk.pk_probe(which, Pp(bad), BLOCKS, ITERS, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None) == 0
            torch.cuda.synchronize()
        res.append(bad.cpu().numpy().astype(np.int64).copy())
    q = [int(res[1][16 * k:16 * k + 16].sum()) for k in range(4)]
    print("%-18s alone: %d wrong   beside the halo convolution: %d wrong  (lanes 0-15 / 16-31 / 32-47 / 48-63: %d / %d / %d / %d)" % (name.value.decode(), int(res[0].sum()), int(res[1].sum()), *q), flush=True)
    which += 1
