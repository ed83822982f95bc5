"""Where a tile's time goes in conv3x3_up2_g1_kernel<HEADS> (de_conv4_0 + heads): s_memtime stamps of consumer wave 0 and producer wave 4 of block 0 over its first 40
tiles — needs a library variant built with -DOMNI_CONV_ABL=32768 (tools/convabl.sh, BITS=32768) given as OMNI_LIB_VARIANT.  Per tile: consumer K loop / heads part /
wait at the barrier; producer load issue / halo arithmetic + LDS writes / wait at the barrier."""
import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
lib = _lib.load()
M, P = int(os.environ.get("M", "144")), 128
net = spherical_fusion(4, 18, (P, P), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
net(torch.rand((1, 3, 512, 1024), device="cuda"))
eng = net._eng; w = eng.w
x = torch.rand((M, P // 2, P // 2, 32), device="cuda"); xs = torch.empty_like(x)
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.omni_sh_from_f32(p(x), p(xs), ctypes.c_size_t(x.numel()), s()), "split")
nb = int(lib.omni_up2_heads_scratch_bytes(M, P)); hs = torch.zeros((nb + 3) // 4, device="cuda")
a, c = torch.empty((M, P, P), device="cuda"), torch.empty((M, P, P), device="cuda")
run = lambda: _lib.check(lib.omni_conv3x3_up2_heads_sh_f16x3(p(xs), p(w["de_conv4_0.w16"]), p(w["de_conv4_0.b"]), p(w["heads.w16f"]), ctypes.c_float(eng.head_bias[0]),
                                                             ctypes.c_float(eng.head_bias[1]), p(hs), ctypes.c_size_t(nb), p(a), p(c), M, P, 1, s()), "up+conv+heads")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
N, K = 40, 5
st = hs[:2 * N * K * 2].cpu().numpy().view(np.int64).reshape(2, N, K)
us = e0.elapsed_time(e1) * 1e3
span = st[0, N - 1, 0] - st[0, 0, 0]
print(f"launch {us:.1f} us (kernel + finish); {N - 1} tiles of block 0 span {span} ticks = {span / (N - 1):.0f} per tile")
print("consumer wave 0: K loop | bias, ReLU, split | heads' matrix instructions | lane exchanges + stores | wait at the barrier")
print("producer wave 4: join hi / lo (waits for the loads) | pixels of cell row 0 | of cell row 1 | address arithmetic + load issue | wait at the barrier   (s_memtime ticks)")
d = lambda w, a, b: st[w, :-1, b] - st[w, :-1, a]
nx = lambda w: st[w, 1:, 0] - st[w, :-1, 4]
for i in range(0, N - 1, 4):
    print(f"{i:3d} | " + " ".join(f"{int(d(0, k, k + 1)[i]):6d}" for k in range(4)) + f" {int(nx(0)[i]):6d} | " + " ".join(f"{int(d(1, k, k + 1)[i]):6d}" for k in range(4)) + f" {int(nx(1)[i]):6d}")
m = lambda v: float(np.mean(v[5:]))
print("mean (tiles 5..): consumer " + " ".join(f"{m(d(0, k, k + 1)):.0f}" for k in range(4)) + f" wait {m(nx(0)):.0f} | producer " + " ".join(f"{m(d(1, k, k + 1)):.0f}" for k in range(4)) + f" wait {m(nx(1)):.0f}")
