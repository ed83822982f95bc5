"""Which CU does block b of a launch land on?  (DEBUG build trace of the pers2equi kernel; prints the CU id sequence of one XCD's blocks)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from omnifusion_amd import _lib as L
lib = L.load_debug()
B, N, P, H, W = 8, 18, 256, 512, 1024
pp = torch.rand((B, N, 1, P, P), device="cuda:0"); erp = torch.empty((B, 1, H, W), device="cuda:0")
trace = torch.zeros((1 << 17, 4), dtype=torch.int64, device="cuda:0")
lib.omni_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
def p2e():
    rc = lib.omni_pers2equi(ctypes.c_void_p(pp.data_ptr()), ctypes.c_void_p(erp.data_ptr()), 0, B, 1, P, P, H, W, 4, ctypes.c_float(80), ctypes.c_float(80), 1, None)
    assert rc == 0
os.environ["OMNI_P2E_DBG"] = "16"
for rep in range(2):
    p2e(); torch.cuda.synchronize(); trace.zero_(); torch.cuda.synchronize(); p2e(); torch.cuda.synchronize()
    t = trace.cpu().numpy()[:4096]
    hw = t[:, 3] & 0xffffffff; xcc = (t[:, 3] >> 32) & 0xff
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; simd = (hw >> 4) & 3; wv = hw & 0xf
    cuid = se * 16 + sh * 8 + cu
    b = np.arange(4096)
    print("xcc of blocks 0..15:", xcc[:16].tolist())
    for x in (0, 3):
        m = b % 8 == x
        ids = cuid[m]
        print(f"XCD-slot {x}: xcc values {sorted(set(xcc[m].tolist()))}; CU id of its blocks 0..79:", ids[:80].tolist())
        u = sorted(set(ids.tolist()))
        print(f"   {len(u)} CUs; per CU block positions (first 3 CUs):", [np.nonzero(ids == c)[0].tolist() for c in u[:3]])
        print("   simd of first 40:", simd[m][:40].tolist())
