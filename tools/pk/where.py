"""Investigation: WHICH pers2equi outputs go wrong beside an MFMA kernel when the library is built with packed fp32 (tools/pk_run.py tools/pk/where.py)."""
import os, sys, ctypes
import numpy as np, torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model._engine import split_weights_f16x3
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
lib = L.load()
Pp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
B, N, P = int(os.environ.get("B", "8")), 18, 128
lay = L.LAYOUT_BNCHW
L.set_option("p2e_gather", int(os.environ.get("GATHER", "1")))
a0 = torch.rand((B, N, 1, P, P), device="cuda")
victim = lambda: pers2equi(a0, 80, 4, P, (512, 1024), None, layout=lay)
ref = victim().clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
M = 144
def sh(t):
    o = torch.empty_like(t); lib.omni_sh_from_f32(Pp(t), Pp(o), ctypes.c_size_t(t.numel()), L.stream_of(t)); return o
x = sh(torch.randn(M, 128, 128, 32, device="cuda")); K = 9 * 32
w16 = split_weights_f16x3(torch.randn(32, K) / np.sqrt(K)).cuda(); b = torch.randn(32, device="cuda")
out = torch.empty(M, 128, 128, 32, device="cuda"); ws = torch.empty(4, device="cuda")
def nz():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.omni_conv2d_sh_f16x3_ws(Pp(x), None, Pp(w16), Pp(b), None, Pp(out), 1, M, 128, 128, 32, 0, 32, 3, 3, 1, 1, 1, 1, Pp(ws), ctypes.c_size_t(16), st) == 0
shown = 0
lanes = np.zeros(64, dtype=np.int64); rows = np.zeros(512, dtype=np.int64); planes = np.zeros(B, dtype=np.int64); nbad = 0
for rep in range(6):
    with torch.cuda.stream(s2):
        for _ in range(40): nz()
    with torch.cuda.stream(s1):
        outs = [victim() for _ in range(12)]
    torch.cuda.synchronize()
    for o in outs:
        d = (o != ref)
        if not bool(d.any()): continue
        nbad += 1
        idx = d.nonzero().cpu().numpy()                     # [k, 4]: b, c, i, j
        np.add.at(lanes, idx[:, 3] % 64, 1); np.add.at(rows, idx[:, 2], 1); np.add.at(planes, idx[:, 0], 1)
        if shown < 3:
            shown += 1
            print("launch with %d wrong values; first ten (plane, row, col, lane): got / want / ratio" % len(idx))
            for b_, c_, i_, j_ in idx[:10]:
                g, w_ = o[b_, c_, i_, j_].item(), ref[b_, c_, i_, j_].item()
                print("   p%d r%3d c%4d l%2d  %.9g / %.9g / %.6f" % (b_, i_, j_, j_ % 64, g, w_, g / w_ if w_ else float("nan")))
            # are whole waves wrong?  (a wave = one row x 64 columns)
            tiles = {}
            for b_, c_, i_, j_ in idx: tiles.setdefault((i_, j_ // 64), set()).add((b_, j_ % 64))
            sizes = sorted(len(v) for v in tiles.values())
            print("   %d (row, 64-column tile) waves touched; wrong values per wave: min %d median %d max %d (of %d = 64 lanes x %d planes)" % (len(tiles), sizes[0], sizes[len(sizes) // 2], sizes[-1], 64 * B, B))
print("wrong launches %d / 72" % nbad)
print("by lane  :", " ".join(str(v) for v in lanes))
print("by plane :", " ".join(str(v) for v in planes))
print("rows with wrong values: %d, first %s" % (int((rows > 0).sum()), np.nonzero(rows)[0][:20]))
