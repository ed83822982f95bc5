"""Per-phase time stamps of the cooperative transformer kernel (block 0): work before each barrier / wait at it.  BS=<panoramas>."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
lib = L.load_debug(); L._lib = lib                   # the DEBUG build (python -m omnifusion_amd.build --debug): the model runs on it too
for name in L.EXPORTS: getattr(lib, name)
lib.omni_up2_heads_scratch_bytes.restype = lib.omni_transformer_scratch_bytes.restype = lib.omni_conv2d_sk_ws_bytes.restype = ctypes.c_size_t
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
B = int(os.environ.get("BS", "1"))
rgb = torch.rand((B, 3, 512, 1024), device="cuda")
for _ in range(5): net(rgb)
buf = torch.zeros(128, dtype=torch.int64, device="cuda")
lib.omni_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
net.LANES = 1
for _ in range(3): net(rgb)
torch.cuda.synchronize()
lib.omni_debug_set_trace(None)
t = buf.cpu().tolist()
names = ["LN1+qkv", "attention", "proj", "LN2+fc1", "fc2"]
tot_w = tot_b = 0
for k in range(1, 31):
    work = (t[2 * k - 1] - t[2 * k - 2]) / 100.0; wait = (t[2 * k] - t[2 * k - 1]) / 100.0
    tot_w += work; tot_b += wait
    if k <= 10: print("layer %d %-10s work %6.2f us   barrier %6.2f us" % ((k - 1) // 5, names[(k - 1) % 5], work, wait))
print("sum over 30 phases: work %.1f us, barrier waits %.1f us; encoder_norm %.2f us; total %.1f us" % (tot_w, tot_b, (t[61] - t[60]) / 100.0, (t[61] - t[0]) / 100.0))
