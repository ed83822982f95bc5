"""VERDICT r5 #4, the measured half of the pricing: Winograd F(2x2, 3x3) on the f16x3 arithmetic at ONE shape (layer3: 3x3, 256 -> 256, 8 x 8 images, 144 patches).
The two transforms run in torch on the GPU (fp32; this is a TOOL, not the product), the multiply stage is the product's own f16x3 GEMM (omni_conv2d_sh_f16x3_ws as a
1 x 1 convolution, fp32 output) over 16 x tiles rows with PER-POSITION weights — sixteen launches of [2304 x 256] . [256 x 256] —, so the numbers are
  (a) ACCURACY: max |d| of the Winograd result against float64 torch, beside the direct f16x3 convolution's on the same data;
  (b) the multiply stage's time as ONE launch over all 16 x 2304 rows (one weight set: same matrix work and traffic) and of one position's GEMM alone, against the direct convolution.
Kill criterion of the review: <= 34 us for the whole layer and |d| <= 5e-6."""
import sys, os, ctypes, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
from omnifusion_amd.model._engine import split_weights_f16x3
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, H, W, C, Co = 144, 8, 8, 256, 256
g = torch.Generator().manual_seed(5)
x = torch.relu(torch.randn(M, H, W, C, generator=g))                   # post-ReLU activations, as layer3 sees them
w = torch.randn(Co, C, 3, 3, generator=g) / np.sqrt(9 * C)
b = torch.randn(Co, generator=g)
ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)          # float64
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def sh(t):
    o = torch.empty_like(t); assert lib.omni_sh_from_f32(P(t), P(o), ctypes.c_size_t(t.numel()), S()) == 0; return o


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); torch.cuda._sleep(int(8e6))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


X = x.cuda()
# ---- direct f16x3 convolution (the product)
XS, W16, B = sh(X), split_weights_f16x3(w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous()).cuda(), b.cuda()
out = torch.empty(M, H, W, Co, device="cuda")
def direct():
    assert lib.omni_conv2d_sh_f16x3_ws(P(XS), None, P(W16), P(B), None, P(out), 0, M, H, W, C, 0, Co, 3, 3, 1, 1, 0, 1, None, ctypes.c_size_t(0), S()) == 0
direct(); torch.cuda.synchronize()
d_direct = (out.cpu().double() - ref).abs().max().item()
# ---- Winograd: input transform (fp32 on the GPU), U = G g G^T in float64 -> fp32 -> split
xp = F.pad(X.permute(0, 3, 1, 2), (1, 1, 1, 1))                        # [M, C, 10, 10]
tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                             # [M, C, 4, 4, 4, 4] (tile row, tile col, 4 x 4 window)
bt = BT.float().cuda()
def in_transform():
    return torch.einsum("ij,mcyxjk,lk->ilmyxc", bt, tiles, bt).reshape(16, M * 16, C).contiguous()     # V[xi*4+nu][tile][c] = B^T d B
V = in_transform()
U = torch.einsum("ij,ocjk,lk->iloc", G, w.double(), G).reshape(16, Co, C)                                # [16][Co][C]
U16 = [split_weights_f16x3(U[p].float().contiguous()).cuda() for p in range(16)]
NT = M * 16
VS = [sh(V[p].reshape(NT, 1, 1, C).contiguous()) for p in range(16)]
Mp = torch.empty(16, NT, Co, device="cuda")
def multiply16():
    for p in range(16):
        assert lib.omni_conv2d_sh_f16x3_ws(P(VS[p]), None, P(U16[p]), None, None, P(Mp[p]), 0, NT, 1, 1, C, 0, Co, 1, 1, 1, 0, 0, 1, None, ctypes.c_size_t(0), S()) == 0
multiply16(); torch.cuda.synchronize()
at = AT.float().cuda()
def out_transform():
    m = Mp.reshape(4, 4, M, 4, 4, Co)
    y = torch.einsum("ij,jkmyxo,lk->myixlo", at, m, at)                # [M, ty, i, tx, l, Co]
    return y.reshape(M, 8, 8, Co) + B
Y = out_transform(); torch.cuda.synchronize()
d_wino = (Y.cpu().double() - ref).abs().max().item()
# the same Winograd in float64 end to end (the algorithm's own error is rounding only) and with fp32 transforms + float64 products (what the transforms cost)
V64 = torch.einsum("ij,mcyxjk,lk->ilmyxc", BT, tiles.cpu().double(), BT).reshape(16, NT, C)
M32 = torch.einsum("ptc,poc->pto", V.cpu().double(), U)               # fp32-rounded V, exact products
y32 = torch.einsum("ij,jkmyxo,lk->myixlo", AT, M32.reshape(4, 4, M, 4, 4, Co), AT).reshape(M, 8, 8, Co) + b.double()
print(f"layer3-shaped convolution, outputs of magnitude {ref.abs().max().item():.2f} (mean |y| {ref.abs().mean().item():.3f}); |M_p| up to {Mp.abs().max().item():.2f}")
print(f"max |d| vs float64: direct f16x3 {d_direct:.3e} | Winograd F(2x2,3x3) on f16x3 GEMMs {d_wino:.3e} | Winograd with fp32-rounded V and exact products {(y32 - ref).abs().max().item():.3e}")
# ---- times
VA, WA = sh(V.reshape(16 * NT, 1, 1, C).contiguous()), U16[0]
MA = torch.empty(16 * NT, Co, device="cuda")
def multiply1():
    assert lib.omni_conv2d_sh_f16x3_ws(P(VA), None, P(WA), None, None, P(MA), 0, 16 * NT, 1, 1, C, 0, Co, 1, 1, 1, 0, 0, 1, None, ctypes.c_size_t(0), S()) == 0
def one_position():
    assert lib.omni_conv2d_sh_f16x3_ws(P(VS[3]), None, P(U16[3]), None, None, P(Mp[3]), 0, NT, 1, 1, C, 0, Co, 1, 1, 1, 0, 0, 1, None, ctypes.c_size_t(0), S()) == 0
t_dir, t_m1, t_p = timeit(direct), timeit(multiply1), timeit(one_position)
print(f"us: direct {t_dir:.1f} | multiply stage as ONE launch over 16 x {NT} rows {t_m1:.1f} | one position's GEMM [{NT} x {C}] . [{C} x {Co}] alone {t_p:.1f} (x 16 = {16 * t_p:.0f}: sixteen short launches are no option)")
print(f"the multiply stage alone is {t_dir / t_m1:.2f}x the direct convolution's speed; with its fp32 products written ({16 * NT * Co * 4 / 1e6:.1f} MB) and V read ({16 * NT * C * 4 / 1e6:.1f} MB)")

# ---- the product's own Winograd path (round 6, experimental): omni_wino_input_sh + omni_conv3x3_wino_sh_f16x3 (multiply stage and output transform in one kernel)
UW = split_weights_f16x3(U.permute(1, 0, 2).reshape(Co, 16 * C).float().contiguous()).cuda()      # [Cout][16 * C], k = p * C + c
VP = torch.empty(16 * NT * C, device="cuda")
assert lib.omni_wino_input_sh(P(XS), P(VP), M, H, W, C, S()) == 0
vb = torch.empty(16 * NT * C, device="cuda"); assert lib.omni_sh_to_f32(P(VP), P(vb), ctypes.c_size_t(vb.numel()), S()) == 0
print(f"input transform kernel vs torch: max |d| {(vb.reshape(16, NT, C) - V).abs().max().item():.2e}")
R = torch.relu(torch.randn(M, H, W, Co, generator=g)).cuda(); RS = sh(R)
ref_r = torch.relu(ref + R.cpu().double())
ws = torch.empty(4 * M * H * W * Co, device="cuda")
for sk in (1, 2, 4):
    o = torch.full((M, H, W, Co), float("nan"), device="cuda")
    def wino(): assert lib.omni_conv3x3_wino_sh_f16x3(P(VP), P(UW), P(B), P(RS), P(o), 1, M, H, W, C, Co, 1, sk, P(ws), ctypes.c_size_t(ws.numel() * 4), S()) == 0, lib.omni_last_error()
    wino(); torch.cuda.synchronize()
    of = torch.empty_like(o); assert lib.omni_sh_to_f32(P(o), P(of), ctypes.c_size_t(o.numel()), S()) == 0
    print(f"product Winograd (+ residual, ReLU, SH out), positions split {sk}: max |d| vs float64 {(of.cpu().double() - ref_r).abs().max().item():.3e} | kernel(s) {timeit(wino):.1f} us")
def intr(): assert lib.omni_wino_input_sh(P(XS), P(VP), M, H, W, C, S()) == 0
o2 = torch.empty(M, H, W, Co, device="cuda")
def direct_r(): assert lib.omni_conv2d_sh_f16x3_ws(P(XS), None, P(W16), P(B), P(RS), P(o2), 1, M, H, W, C, 0, Co, 3, 3, 1, 1, 1, 1, None, ctypes.c_size_t(0), S()) == 0
print(f"input transform kernel {timeit(intr):.1f} us | direct convolution (+ residual, ReLU, SH out) {timeit(direct_r):.1f} us")
