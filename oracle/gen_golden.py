"""Generate tests/golden/*.npz from the reference ITSELF (TEST INFRASTRUCTURE ONLY).

Runs only in the build container, where /root/reference is mounted: it imports the
reference's Python (oracle/ref_loader.py), feeds it seeded inputs and stores
inputs + outputs as small .npz fixtures.  Fixtures are data (inputs and expected
outputs); no reference source travels.

    python -m oracle.gen_golden            # geometry fixtures G1-G5, G8
    python -m oracle.gen_golden --model    # + full-model fixtures G6, G7 (slow)

Fixture index:
  G1  equi2pers  ERP 64x128  C=3 B=2 nrows=4 P=16
  G2  equi2pers  ERP 128x256 C=3 B=2 nrows=6 P=32 ;  G2b nrows 3 and 5 (ERP 64x128, P=16)
  G2c equi2pers  non-square patch (12,20), scalar fov, C=1
  G2d equi2pers  odd patch sizes (9,9) [quirk q4: 0/0 at the centre] and (9,15)
  G3  pers2equi  [2,1,16,16,18] -> [2,1,64,128] ;  G3b nrows=3 (uncovered pixels, +-59.6 centres), nrows=5
  G4  pers2equi  nrows=6 [2,2,32,32,46] -> [2,2,128,256]
  G5  pers2equi tables x0,y0,x1,y1,mask,w_list at 32x64 / P=8 for nrows 3,4,5,6
  G8  known-answer scalars at BASELINE config 1 scale (512x1024, 18 x 256^2)
  G6  single-pass model, P=128, ERP 64x128, deterministic weights (omnifusion_amd.weights)
  G7  iterative model, iter=2, confidence False/True
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_loader import load_reference, ref_equi2pers, ref_pers2equi, scratch_cwd  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def rng_uniform(seed, shape):
    return np.random.default_rng(seed).random(shape, dtype=np.float32)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def g_equi2pers(name, seed, B, C, H, W, nrows, P, fov):
    erp = rng_uniform(seed, (B, C, H, W))
    pers, xyz, uv, cp = ref_equi2pers(torch.from_numpy(erp), fov, nrows, P)
    save(name, erp=erp, pers=pers.numpy(), xyz=xyz.numpy(), uv=uv.numpy(), center_p=cp.numpy(),
         fov=np.array(fov if isinstance(fov, tuple) else (fov, fov), np.float32),
         nrows=np.int32(nrows), patch=np.array(P if isinstance(P, tuple) else (P, P), np.int32))


def g_pers2equi(name, seed, B, C, P, N, nrows, H, W, fov=(80, 80)):
    pers = rng_uniform(seed, (B, C, P, P, N))
    erp = ref_pers2equi(torch.from_numpy(pers), fov, nrows, (P, P), (H, W))
    save(name, pers=pers, erp=erp.numpy(), fov=np.array(fov, np.float32), nrows=np.int32(nrows),
         erp_size=np.array((H, W), np.int32))


def g_tables(name, nrows, P, H, W, fov=(80, 80)):
    """G5: the reference's own cached table (pers2equi_v3.py:155) at tiny size."""
    ref = load_reference()
    N = {3: 10, 4: 18, 5: 26, 6: 46}[nrows]
    with scratch_cwd():
        ref.pers2equi(torch.zeros(1, 1, P, P, N), fov, nrows, (P, P), (H, W), "tbl")
        t = torch.load(os.path.join("grid", "tbl.pth"))
    save(name, x0=t["x0"].numpy().astype(np.int32), y0=t["y0"].numpy().astype(np.int32),
         x1=t["x1"].numpy().astype(np.int32), y1=t["y1"].numpy().astype(np.int32),
         mask=t["mask"].numpy().astype(np.int32), w_list=t["w_list"].numpy(),
         nrows=np.int32(nrows), patch=np.int32(P), erp_size=np.array((H, W), np.int32),
         fov=np.array(fov, np.float32))


def g_known_answers():
    """G8: config-scale scalars (inputs regenerated from numpy seeds in the test)."""
    erp = rng_uniform(100, (1, 3, 512, 1024))
    pers, xyz, uv, cp = ref_equi2pers(torch.from_numpy(erp), (80, 80), 4, (256, 256))
    pin = rng_uniform(101, (1, 1, 256, 256, 18))
    e = ref_pers2equi(torch.from_numpy(pin), (80, 80), 4, (256, 256), (512, 1024))
    # a strided sub-sample keeps spatial information without a multi-MB fixture
    ka = dict(
        pers_sum=float(pers.double().sum()), erp_sum=float(e.double().sum()),
        pers_abs_sum=float(pers.double().abs().sum()),
    )
    save("G8_config1", pers_sub=pers.numpy()[:, :, ::8, ::8, :], xyz_sub=xyz.numpy()[:, :, ::8, ::8],
         uv_sub=uv.numpy()[:, :, ::8, ::8], center_p=cp.numpy(), erp_sub=e.numpy()[:, :, ::4, ::4],
         erp_rows=e.numpy()[:, :, [0, 1, 255, 256, 510, 511], :])
    with open(os.path.join(OUT, "G8_config1.json"), "w") as f:
        json.dump(ka, f, indent=1)
    # nrows=6 at 1024x2048 (cfg 3) pers2equi + equi2pers sub-samples
    erp3 = rng_uniform(102, (1, 1, 1024, 2048))
    p3, _, _, _ = ref_equi2pers(torch.from_numpy(erp3), (80, 80), 6, (256, 256))
    pin3 = rng_uniform(103, (1, 1, 256, 256, 46))
    e3 = ref_pers2equi(torch.from_numpy(pin3), (80, 80), 6, (256, 256), (1024, 2048))
    # The reference emits NaN where cos_c of some patch is EXACTLY 0 in its fp32 arithmetic
    # (X = +-inf, inf * mask(0) = NaN, pers2equi_v3.py:113,144 -> the whole pixel after :192);
    # record where that happens at this size.
    e3n = e3.numpy()
    save("G8_config3", pers_sub=p3.numpy()[:, :, ::8, ::8, :], erp_sub=e3n[:, :, ::8, ::8],
         pers_sum=np.float64(p3.double().sum()), erp_sum=np.float64(np.nansum(e3n.astype(np.float64))),
         erp_nan_idx=np.argwhere(~np.isfinite(e3n)).astype(np.int32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", action="store_true")
    ap.add_argument("--only-model", action="store_true")
    ap.add_argument("--only-known", action="store_true")
    args = ap.parse_args()
    if args.only_known:
        g_known_answers()
        return
    if not args.only_model:
        g_equi2pers("G1_equi2pers_n4", 1, 2, 3, 64, 128, 4, 16, (80, 80))
        g_equi2pers("G2_equi2pers_n6", 2, 2, 3, 128, 256, 6, 32, (80, 80))
        g_equi2pers("G2b_equi2pers_n3", 3, 1, 2, 64, 128, 3, 16, (80, 80))
        g_equi2pers("G2b_equi2pers_n5", 4, 1, 2, 64, 128, 5, 16, (80, 80))
        g_equi2pers("G2c_equi2pers_rect", 5, 1, 1, 48, 80, 4, (12, 20), (60, 90))
        g_equi2pers("G2d_equi2pers_odd", 6, 1, 1, 37, 91, 4, (9, 9), (80, 80))       # quirk q4: NaN centre rays
        g_equi2pers("G2d_equi2pers_odd2", 7, 1, 1, 37, 91, 5, (9, 15), (80, 80))    # linspace(0,1,15)[7] != 0.5
        g_pers2equi("G3_pers2equi_n4", 11, 2, 1, 16, 18, 4, 64, 128)
        g_pers2equi("G3b_pers2equi_n3", 12, 1, 1, 16, 10, 3, 64, 128)
        g_pers2equi("G3b_pers2equi_n5", 13, 1, 1, 16, 26, 5, 64, 128)
        g_pers2equi("G4_pers2equi_n6", 14, 2, 2, 32, 46, 6, 128, 256)
        g_pers2equi("G4b_pers2equi_fov", 15, 1, 1, 24, 18, 4, 48, 96, fov=(60, 100))
        for nr in (3, 4, 5, 6):
            g_tables(f"G5_tables_n{nr}", nr, 8, 32, 64)
        g_known_answers()
    if args.model or args.only_model:
        from oracle.gen_golden_model import main as model_main
        model_main()


if __name__ == "__main__":
    main()
