"""Ablation of the equi2pers box kernel (DEBUG build).  OMNI_E2P_DBG bits: 1 no stores, 2 no DMA."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
lib = L.load_debug()
B, N, P, H, W = int(os.environ.get("B", "8")), 18, int(os.environ.get("P", "256")), 512, 1024
x = torch.rand((B, 3, H, W), device="cuda:0"); out = torch.empty((B, N, 3, P, P), device="cuda:0")
def run():
    rc = lib.omni_equi2pers(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), 0, B, 3, H, W, P, P, 4, ctypes.c_float(80), ctypes.c_float(80), 1, None)
    assert rc == 0, lib.omni_last_error()
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"OMNI_E2P_DBG={os.environ.get('OMNI_E2P_DBG', '0'):>2s} B={B} P={P}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
