import sys, ctypes, torch
sys.path.insert(0, '.')
from omnifusion_amd import _lib
lib = _lib.load_debug()
n = 18 * 256 * 256          # one "plane" = all patches of one (b,c)
for planes in (1, 24):
    buf = torch.empty(n * planes, device="cuda")
    for mode in (0, 1, 2):
        def run():
            lib.omni_debug_fill(ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(n), planes, mode, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        print(f"planes={planes} mode={mode}: {t*1e6:.1f} us  {n*planes*4/t/1e9:.0f} GB/s")
