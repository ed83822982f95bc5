"""Where a block's time goes in conv3x3_halo_sh_kernel (layer1 / layer2 shapes): s_memtime stamps of wave 0 of blocks 0 and 600 — needs a library variant built with
-DOMNI_CONV_ABL=32768 (BITS=32768 tools/convabl.sh) given as OMNI_LIB_VARIANT.  ONLY=<row of tools/convbench.py's table> (default 0 = layer1)."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
from omnifusion_amd.model._engine import split_weights_f16x3
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("M", "144"))
CFGS = {"layer1": (32, 32, 64, 64, True), "layer2": (16, 16, 128, 128, True), "de2_1": (32, 32, 128, 64, False)}
name = os.environ.get("LAYER", "layer1")
H, W, C1, Cout, use_res = CFGS[name]
x1 = torch.randn(M, H, W, C1, device="cuda"); K = C1 * 9
w16 = split_weights_f16x3(torch.randn(Cout, K) / np.sqrt(K)).cuda(); b = torch.randn(Cout, device="cuda")
res = torch.randn(M, H, W, Cout, device="cuda") if use_res else None
out = torch.empty(M, H, W, Cout, device="cuda"); ws = torch.zeros(64 << 20, device="cuda")
def sh(t):
    if t is None: return None
    o = torch.empty_like(t); lib.omni_sh_from_f32(P(t), P(o), ctypes.c_size_t(t.numel()), S()); return o
x1, res = sh(x1), sh(res)
def run():
    rc = lib.omni_conv2d_sh_f16x3_ws(P(x1), None, P(w16), P(b), P(res), P(out), 1, M, H, W, C1, 0, Cout, 3, 3, 1, 1, 1, 1, P(ws), ctypes.c_size_t(ws.numel() * 4), S())
    assert rc == 0, lib.omni_last_error()
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
st = ws[:256].cpu().numpy().view(np.int64).reshape(2, 64)
print(f"{name} M={M}: launch {e0.elapsed_time(e1) * 1e3:.1f} us (with the stamps in)")
G = C1 // 32
for bi, blk in enumerate((0, 600)):
    s = st[bi]; t0 = s[0]
    print(f"block {blk}: prologue: index arithmetic {s[63] - t0} + first halo / weights issued {s[1] - s[63]} ticks")
    print("   stage | waits + barrier | next weights issued | reads + matrix instructions issued | gap to the next stage")
    for k in range(3 * G):
        b0, b1, b2, b3 = s[4 + 4 * k: 8 + 4 * k]
        nxt = s[4 + 4 * (k + 1)] if k + 1 < 3 * G else s[2]
        print(f"   {k:5d} | {b1 - b0:8d} | {b2 - b1:8d} | {b3 - b2:8d} | {nxt - b3:8d}")
    print(f"   epilogue {s[62] - s[2]} ticks to its last store issued + {s[3] - s[62]} until the stores are acknowledged; whole block {s[3] - t0} ticks ({9 * G * 2 * 2 * 3} matrix instructions per wave = {9 * G * 2 * 2 * 3 * 32} pipe cycles)")
