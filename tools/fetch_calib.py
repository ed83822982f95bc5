"""FETCH_SIZE calibration kernels (debug build: omni_debug_calib) — run under `rocprofv3 --pmc FETCH_SIZE`, see tools/fetch_calib.sh.
Each launch reads BYTES bytes of a fresh 96-MB buffer ONCE (three buffers in rotation: nothing is found in the 32 MB of L2)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
lib = L.load_debug()
BYTES = 96 << 20
bufs = [torch.rand(BYTES // 4, device="cuda") for _ in range(3)]
sink = torch.zeros(4, device="cuda")
S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    for pat, param in ((0, 0), (1, 0), (2, 0), (3, 0), (4, 32), (4, 64), (4, 128), (4, 256)):
        rc = lib.omni_debug_calib(pat, ctypes.c_void_p(bufs[rep].data_ptr()), ctypes.c_size_t(BYTES), param, ctypes.c_void_p(sink.data_ptr()), S)
        assert rc == 0, lib.omni_last_error()
        torch.cuda.synchronize()
print("done", BYTES)
