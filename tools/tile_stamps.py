"""Where a block's time goes in conv_sh_kernel (layer3 / layer4 / de_conv0_0 shapes; loader waves): s_memtime stamps of matrix wave 0 and of the first loader wave of
block 0 over its first 28 K steps — needs a library variant built with -DOMNI_CONV_ABL=32768 (BITS=32768 tools/convabl.sh) given as OMNI_LIB_VARIANT.  LAYER=layer3 | de0_0"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
from omnifusion_amd.model._engine import split_weights_f16x3
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("M", "144"))
CFGS = {"layer3": (8, 8, 256, 256, True), "de0_0": (8, 8, 512, 256, False)}
name = os.environ.get("LAYER", "layer3")
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
s = ws[:512].cpu().numpy().view(np.int64)
ksteps = 9 * C1 // 32
print(f"{name} M={M}: launch {e0.elapsed_time(e1) * 1e3:.1f} us (with the stamps in); {ksteps} K steps per block")
print(f"block 0: start -> first K step {s[8] - s[0]} ticks | K loop {s[1] - s[8]} | epilogue {s[2] - s[1]} | whole block {s[2] - s[0]}")
print("  step | matrix wave 0: wait + barrier, reads + matrix instructions | loader wave: wait for its pieces, barrier, issue of the next stage, idle until the next step")
for k in range(0, 27):
    m0, m1, m2 = s[8 + 4 * k], s[9 + 4 * k], s[8 + 4 * (k + 1)]
    l0, l1, l2, l3, l4 = s[128 + 4 * k], s[129 + 4 * k], s[130 + 4 * k], s[131 + 4 * k], s[128 + 4 * (k + 1)]
    print(f"  {k:4d} | {m1 - m0:6d} {m2 - m1:6d} | {l1 - l0:6d} {l2 - l1:6d} {l3 - l2:6d} {l4 - l3:6d}")
mm = lambda a: float(np.mean(a[4:]))
M0 = np.array([s[9 + 4 * k] - s[8 + 4 * k] for k in range(27)]); M1 = np.array([s[8 + 4 * (k + 1)] - s[9 + 4 * k] for k in range(27)])
L = [np.array([s[128 + 4 * k + j + 1] - s[128 + 4 * k + j] for k in range(27)]) for j in range(4)]
print(f"mean (steps 4..): matrix wave wait + barrier {mm(M0):.0f}, reads + matrix instructions {mm(M1):.0f} (12 matrix instructions = 384 pipe cycles; two matrix waves per SIMD) | loader: wait {mm(L[0]):.0f}, barrier {mm(L[1]):.0f}, issue {mm(L[2]):.0f}, to next step {mm(L[3]):.0f}")

if os.environ.get("OMNI_CONV_PINGPONG", "1") != "0":
    # the ping-pong schedule (conv_sh_kernel<.., PP>): matrix wave 0 = group A: reads(k) | wait at a_k | its 12 matrix instructions | wait at b_k
    R = np.array([s[11 + 4 * k] - s[8 + 4 * k] for k in range(27)]); WA = np.array([s[9 + 4 * k] - s[11 + 4 * k] for k in range(27)])
    MI = np.array([s[10 + 4 * k] - s[9 + 4 * k] for k in range(27)]); WB = np.array([s[8 + 4 * (k + 1)] - s[10 + 4 * k] for k in range(27)])
    print(f"ping-pong, group A wave 0 (mean over steps 4..): fragment reads + lgkmcnt(0) {mm(R):.0f} | at barrier a_k {mm(WA):.0f} | 12 matrix instructions issued {mm(MI):.0f} | at barrier b_k {mm(WB):.0f} | K step {mm(R) + mm(WA) + mm(MI) + mm(WB):.0f}")
