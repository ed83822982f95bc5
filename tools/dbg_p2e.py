import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
from omnifusion_amd import _lib as L
from oracle import c_oracle as co
NP = {3: 10, 4: 18, 5: 26, 6: 46}
for nrows in (4, 5, 6):
    for (B, C) in ((1, 1), (2, 1), (1, 2), (2, 2), (1, 3), (3, 3)):
        N = NP[nrows]; P = 32; H, W = 128, 256
        x = np.random.default_rng(1).random((B, C, P, P, N), dtype=np.float32)
        ref, cover = co.pers2equi(x, (80, 80), nrows, (P, P), (H, W), want_cover=True)
        xt = torch.from_numpy(x).cuda()
        g1 = pers2equi(xt, (80, 80), nrows, (P, P), (H, W), "a").cpu().numpy()
        g2 = pers2equi(xt.permute(0, 4, 1, 2, 3).contiguous(), (80, 80), nrows, (P, P), (H, W), "a", layout=L.LAYOUT_BNCHW).cpu().numpy()
        d1 = np.abs(g1 - ref); d2 = np.abs(g2 - ref)
        bad = (d1 > 2e-4)
        print(nrows, B, C, "ref-layout bad", int(bad.sum()), "max", d1.max(), "| planar bad", int((d2 > 2e-4).sum()), d2.max(),
              "| cover at bad", np.unique(cover[bad.any(axis=(0, 1))]) if bad.any() else "")
