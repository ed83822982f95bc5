"""MFMA issue-pattern microbenchmark (debug library): cycles per v_mfma_f32_32x32x16_f16 by accumulator dependency pattern."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib
lib = _lib.load_debug()
cyc = torch.zeros(1, dtype=torch.int64, device="cuda"); sink = torch.zeros(4, device="cuda")
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {0: "one chain", 1: "acc1,acc,acc1", 2: "two f16x3 tiles interleaved", 3: "four chains", 4: "acc,acc1,acc1"}
per = {0: 18, 1: 18, 2: 36, 3: 24, 4: 18}
for threads in (256, 512):
    for blocks in (1, 256):
        for pat in range(5):
            iters = 2000
            for _ in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.omni_debug_mfma_pattern(pat, iters, blocks, threads, ctypes.c_void_p(cyc.data_ptr()), ctypes.c_void_p(sink.data_ptr()), S())
                e1.record(); torch.cuda.synchronize()
            n = iters * per[pat]
            ms = e0.elapsed_time(e1)
            print("threads %d blocks %3d %-30s %6.1f ticks/MFMA (s_memtime)  %6.1f ns/MFMA wall -> %.2f GHz-cycles@32" % (
                threads, blocks, names[pat], cyc.item() / n, ms * 1e6 / n, 0), flush=True)
