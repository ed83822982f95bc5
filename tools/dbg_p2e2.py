import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
from omnifusion_amd import _lib as L
from oracle import c_oracle as co
nrows, N, P, H, W = 6, 46, 32, 128, 256
x = np.random.default_rng(1).random((1, 1, P, P, N), dtype=np.float32)
ref = co.pers2equi(x, (80, 80), nrows, (P, P), (H, W))
xt = torch.from_numpy(x).cuda()
g1 = pers2equi(xt, (80, 80), nrows, (P, P), (H, W), "a").cpu().numpy()
g1b = pers2equi(xt, (80, 80), nrows, (P, P), (H, W), "a").cpu().numpy()
print("deterministic:", np.array_equal(g1, g1b))
bad = np.abs(g1 - ref)[0, 0] > 2e-4
print("bad per row:", [(i, int(bad[i].sum())) for i in range(H) if bad[i].any()])
# single-patch probes: only patch k nonzero
t = co.pers2equi_tables((80, 80), nrows, (P, P), (H, W))
for k in (0, 10, 31, 32, 33, 40, 45):
    xk = np.zeros_like(x); xk[..., k] = 1.0
    r = co.pers2equi(xk, (80, 80), nrows, (P, P), (H, W))
    g = pers2equi(torch.from_numpy(xk).cuda(), (80, 80), nrows, (P, P), (H, W), "a").cpu().numpy()
    print("patch", k, "max diff", np.abs(g - r).max(), "ref nonzero px", int((r > 0).sum()), "gpu nonzero", int((g > 0).sum()))
