"""Full-model golden fixtures G6/G7 from the reference itself (TEST INFRASTRUCTURE ONLY; build
container only).  Weights: omnifusion_amd.weights.make_state_dict(seed=42) loaded into the
reference modules; inputs: numpy-seeded smooth panoramas 64x128 (ERP size is independent of the
patch size, so the fixtures stay small while the whole 35.7-GMAC network at P=128 is exercised).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.ref_loader import load_reference, scratch_cwd  # noqa: E402
from omnifusion_amd.weights import make_state_dict  # noqa: E402
from _util import smooth_erp  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    ref = load_reference()
    torch.manual_seed(0)
    rgb = smooth_erp(600, 2, 3, 64, 128, k=9, passes=1)
    x = torch.from_numpy(rgb)
    # ---- G6 single pass
    net = ref.spherical_fusion(nrows=4, npatches=18, patch_size=(128, 128), fov=(80, 80))
    net.load_state_dict(make_state_dict(42, 18, False))
    net.eval()
    taps = {}
    hooks = [net.de_conv4_0.register_forward_hook(lambda m, i, o: taps.__setitem__("de_conv4_0", o.detach().clone())),
             net.pred.register_forward_hook(lambda m, i, o: taps.__setitem__("pred_raw", o.detach().clone())),
             net.weight_pred.register_forward_hook(lambda m, i, o: taps.__setitem__("weight_raw", o.detach().clone())),
             net.layer4.register_forward_hook(lambda m, i, o: taps.__setitem__("layer4_pre", o.detach().clone()))]
    with scratch_cwd(), torch.no_grad():
        out_conf = net(x, confidence=True)
        out_noconf = net(x, confidence=False)
    for h in hooks:
        h.remove()
    _save("G6_model_single", rgb=rgb, depth_conf=out_conf.numpy(), depth_noconf=out_noconf.numpy(),
          pred_sub=torch.relu(taps["pred_raw"]).numpy()[:, :, ::4, ::4, :],
          weight_sub=torch.sigmoid(taps["weight_raw"]).numpy()[:, :, ::4, ::4, :],
          de_conv4_0_sub=taps["de_conv4_0"].numpy()[:, :, ::8, ::8, :],
          layer4_pre=taps["layer4_pre"].numpy())
    # ---- G6b: BASELINE config-1 ERP size (512x1024), P=128, B=1; the input is tests/_util.smooth_erp(77, ...)
    rgb512 = torch.from_numpy(smooth_erp(77, 1, 3, 512, 1024))
    with scratch_cwd(), torch.no_grad():
        o512 = net(rgb512, confidence=True)
    _save("G6b_model_single_512x1024", depth_conf_sub=o512.numpy()[:, :, ::2, ::2])
    # ---- G7 iterative, iter=2, confidence False (as test.py:198 calls it) and True
    neti = ref.spherical_fusion_iterative(nrows=4, npatches=18, patch_size=(128, 128), fov=(80, 80))
    neti.load_state_dict(make_state_dict(42, 18, True))
    neti.eval()
    with scratch_cwd(), torch.no_grad():
        o_f = neti(x[:1], 2, confidence=False)
    with scratch_cwd(), torch.no_grad():
        o_t = neti(x[:1], 2, confidence=True)
    _save("G7_model_iterative", rgb=rgb[:1], it0=o_f[0].numpy(), it1=o_f[1].numpy(),
          it0_conf=o_t[0].numpy(), it1_conf=o_t[1].numpy())
    # ---- G7b: nrows = 6 (46 patches, the BASELINE config-3 geometry) at a small ERP, iterative iter = 2
    rgb6 = smooth_erp(601, 1, 3, 96, 192, k=9, passes=1)
    net6 = ref.spherical_fusion_iterative(nrows=6, npatches=46, patch_size=(128, 128), fov=(80, 80))
    net6.load_state_dict(make_state_dict(42, 46, True))
    net6.eval()
    with scratch_cwd(), torch.no_grad():
        o6 = net6(torch.from_numpy(rgb6), 2, confidence=False)
    _save("G7b_model_iterative_n6", rgb=rgb6, it0=o6[0].numpy(), it1=o6[1].numpy())


def main_presets():
    """G6c / G7c (round 4): the two `nrows` presets the full network had no reference-backed check at — nrows 3 (10 patches, the
    preset whose pers2equi centres differ from equi2pers's, q7, and which leaves ERP pixels uncovered) and nrows 5 (26 patches), equi2pers_v3.py:40-47.
    Single pass with and without confidence, and the 2-iteration iterative model; same weight generator, new files only."""
    ref = load_reference()
    for nrows, N, seed in ((3, 10, 602), (5, 26, 603)):
        torch.manual_seed(0)
        rgb = smooth_erp(seed, 2, 3, 64, 128, k=9, passes=1)
        x = torch.from_numpy(rgb)
        net = ref.spherical_fusion(nrows=nrows, npatches=N, patch_size=(128, 128), fov=(80, 80))
        net.load_state_dict(make_state_dict(42, N, False))
        net.eval()
        with scratch_cwd(), torch.no_grad():
            oc = net(x, confidence=True)
        with scratch_cwd(), torch.no_grad():
            on = net(x, confidence=False)
        neti = ref.spherical_fusion_iterative(nrows=nrows, npatches=N, patch_size=(128, 128), fov=(80, 80))
        neti.load_state_dict(make_state_dict(42, N, True))
        neti.eval()
        with scratch_cwd(), torch.no_grad():
            oi = neti(x[:1], 2, confidence=False)
        _save(f"G6c_model_n{nrows}", rgb=rgb, depth_conf=oc.numpy(), depth_noconf=on.numpy(), it0=oi[0].numpy(), it1=oi[1].numpy())


if __name__ == "__main__":
    if "--presets" in sys.argv:
        main_presets()
    else:
        main()
