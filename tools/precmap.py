"""Per-layer precision map of the f16x3 network (debug library): for every convolution / GEMM launched through the SH
kernels, drop the weight-lo term (A_hi.W_lo), the activation-lo term (A_lo.W_hi) or both IN THAT LAYER ONLY and record the
max abs deviation of the final depth from the reference golden G6 (gate: 1e-3)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib
_lib.load = _lib.load_debug
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.model import _engine
from omnifusion_amd.weights import make_state_dict
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G6_model_single.npz"))
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
rgb = torch.from_numpy(g["rgb"]).cuda()
ref = g["depth_conf"]
keys = []
target = {"key": None, "bits": 0}
orig_conv, orig_gemm = _engine.Engine._conv, _engine.Engine._gemm_sh
def conv(self, x, key, *a, **k):
    if key not in keys: keys.append(key)
    os.environ["OMNI_CONV_DBG"] = str(target["bits"] if (target["key"] == key or target["key"] == "*") else 0)
    try: return orig_conv(self, x, key, *a, **k)
    finally: os.environ["OMNI_CONV_DBG"] = "0"
def gemm(self, x, w16key, *a, **k):
    if w16key not in keys: keys.append(w16key)
    os.environ["OMNI_CONV_DBG"] = str(target["bits"] if (target["key"] == w16key or target["key"] == "*") else 0)
    try: return orig_gemm(self, x, w16key, *a, **k)
    finally: os.environ["OMNI_CONV_DBG"] = "0"
_engine.Engine._conv, _engine.Engine._gemm_sh = conv, gemm
def err():
    return float(np.abs(net(rgb, confidence=True).cpu().numpy() - ref).max())
base = err()
print("baseline (all three terms everywhere): max |d| vs reference = %.3g" % base)
for bits, name in ((16, "no W_lo"), (32, "no A_lo"), (48, "fp16 x fp16")):
    target.update(key="*", bits=bits); print("ALL layers %-12s: %.3g" % (name, err()))
print("%-28s %10s %10s %10s" % ("layer", "no W_lo", "no A_lo", "fp16xfp16"))
rows = []
for key in list(keys):
    r = []
    for bits in (16, 32, 48):
        target.update(key=key, bits=bits); r.append(err())
    rows.append((key, r)); print("%-28s %10.3g %10.3g %10.3g" % (key, *r), flush=True)
ok = [k for k, r in rows if r[2] <= 3e-4]
print("layers that could run fp16 x fp16 alone with max|d| <= 3e-4:", len(ok), "of", len(rows), "(the transformer GEMMs)" if all(k.startswith("t") for k in ok) else ok)
# all of them together, and what it buys at batch 1 (the debug library's run-time branches cost a little themselves)
import time
class _Set(str):
    def __eq__(self, other): return other in ok
    __hash__ = str.__hash__
for bits, name in ((0, "three terms"), (48, "fp16 x fp16 in those layers")):
    target.update(key=_Set("set"), bits=bits)
    e = err()
    one = torch.rand((1, 3, 512, 1024), device="cuda")
    for _ in range(5): net(one, confidence=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): net(one, confidence=True)
    torch.cuda.synchronize()
    print("%-30s: max |d| = %.3g, batch-1 forward %.3f ms" % (name, e, (time.perf_counter() - t0) / 30 * 1e3))
