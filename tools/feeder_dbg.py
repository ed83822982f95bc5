import sys, os, numpy as np, torch, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from _util import smooth_erp
from omnifusion_amd import _lib as L
from omnifusion_amd.data import DeviceFeeder, preprocess_rgb
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
DEV = "cuda:0"
lib = L.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
sd = make_state_dict(42, 18, False)
if mode in ("all", "cap"):
    net0 = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net0.load_state_dict(sd)
    rgb0 = torch.from_numpy(smooth_erp(41, 1, 3, 64, 128)).to(DEV)
    ref = net0(rgb0).clone()
    lib.omni_geometry_cache_clear(); L.set_option("geom_cache_max", 2)
    run0 = net0.graphed(rgb0)
    for k in range(6): equi2pers_patches(torch.rand((1, 1, 40 + 2 * k, 96), device=DEV), 80, 4, 8)
    torch.cuda.synchronize(); assert torch.equal(run0(rgb0), ref)
    L.set_option("geom_cache_max", 16); lib.omni_geometry_cache_clear()
if mode in ("all", "ovf"):
    net1 = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net1.load_state_dict(sd)
    rgb1 = torch.from_numpy(smooth_erp(43, 1, 3, 64, 128)).to(DEV)
    net1(rgb1); print("ovf0", net1.overflowed())
    net1.load_state_dict({k: (v * 4e3 if k == "conv1.weight" else v) for k, v in sd.items()})
    o = net1(rgb1); print("finite", bool(torch.isfinite(o).all()), "ovf", net1.overflowed(), net1.overflowed())
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(sd)
rng = np.random.default_rng(10)
batches = [torch.from_numpy(rng.integers(0, 256, (2, 128, 256, 3), dtype=np.uint8)).pin_memory() for _ in range(9)]
want = [net(preprocess_rgb(b.cuda(), (128, 256))).clone() for b in batches]
for rep in range(6):
    run = net.pipelined(3)
    feeder = DeviceFeeder(batches, (128, 256), depth=3, out_buffers=2)
    pend = []
    for rgb in feeder:
        p = run(rgb); feeder.done_with(rgb, p.input_read); pend.append(p)
    got = [p.get() for p in pend]
    torch.cuda.synchronize()
    d = [float((g - w).abs().max()) for g, w in zip(got, want)]
    nanw = [bool(torch.isnan(w).any()) for w in want]
    print(mode, "rep", rep, "max diffs", ["%.2g" % v for v in d], "want has nan", any(nanw))
