"""Cost of a device-wide barrier between co-resident blocks (agent-scope counter + spin), with and without a 16-byte-per-thread exchange
through device-scope stores / loads — what a cooperative (one-launch) transformer would pay per GEMM boundary.  DEBUG build."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnifusion_amd import _lib as L
lib = L.load_debug()
P = lambda t: ctypes.c_void_p(t.data_ptr())
iters = 200
for blocks in (16, 64, 128, 256):
    for threads in (256, 512):
        for payload in (0, 1):
            cnt = torch.zeros(iters, dtype=torch.int32, device="cuda"); buf = torch.zeros(2 * blocks * threads * 4, device="cuda")
            cyc = torch.zeros(1, dtype=torch.int64, device="cuda"); sink = torch.zeros(1, device="cuda")
            for rep in range(2):
                cnt.zero_(); torch.cuda.synchronize()
                rc = lib.omni_debug_grid_barrier(P(cnt), P(buf), iters, blocks, threads, payload, P(cyc), P(sink), None)
                assert rc == 0, lib.omni_last_error()
                torch.cuda.synchronize()
            print(f"{blocks:4d} blocks x {threads:4d} threads, {'16-byte exchange + ' if payload else ''}barrier: {cyc.item() / 100.0 / iters:6.2f} us each")
