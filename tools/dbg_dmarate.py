import sys, ctypes, torch
sys.path.insert(0, '.')
from omnifusion_amd import _lib
lib = _lib.load_debug()
src = torch.randn(1 << 20, device="cuda")          # 4 MiB: L2 / MALL resident
sink = torch.zeros(4, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
for ldskb, name in ((80, "1-2 blocks/CU (80 KB)"), (48, "3 blocks/CU"), (32, "5 blocks/CU")):
    for blocks in (256, 512, 768, 1280):
        iters = 400
        def run(): lib.omni_debug_dma_rate(P(src), src.numel() * 4, iters, blocks, ldskb, P(sink), None)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3
        byts = blocks * iters * 4 * 8192
        print(f"{name:24s} blocks={blocks:5d}: {t*1e6:8.1f} us  {byts/t/1e12:6.2f} TB/s  = {byts/t/256/2.1e9:6.1f} B/clk/CU (at 2.1 GHz)")
