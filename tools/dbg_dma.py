import sys, ctypes, torch
sys.path.insert(0, '.')
from omnifusion_amd import _lib
lib = _lib.load_debug()
src = torch.arange(1024, dtype=torch.float32, device="cuda") + 1
offs = torch.arange(64, dtype=torch.int32) * 16
offs[5] = -2147483648; offs[9] = 4096; offs[17] = 1 << 30; offs[63] = 4080
offs = offs.cuda()
out = torch.zeros(256, device="cuda")
rc = lib.omni_debug_dma_probe(ctypes.c_void_p(src.data_ptr()), 4096, ctypes.c_void_p(offs.data_ptr()), ctypes.c_void_p(out.data_ptr()), None)
torch.cuda.synchronize()
o = out.cpu().reshape(64, 4)
for l in (0, 4, 5, 6, 9, 17, 62, 63): print(l, o[l].tolist())
