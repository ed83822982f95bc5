"""The resample pair at 18x256^2, B=8: one after the other on one stream vs the two operators on two streams (as consecutive pipelined
forwards run them: equi2pers of batch k+1 beside pers2equi of batch k)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnifusion_amd import _lib as L
from omnifusion_amd.model.spherical_model import _concurrent_streams
from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers_patches
from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi
B, P, N = 8, 256, 18
lay = L.LAYOUT_BNCHW
rgb = torch.rand((B, 3, 512, 1024), device="cuda"); dp = torch.rand((B, N, 1, P, P), device="cuda")
nbytes = B * 3 * (512 * 1024 + P * P * N) * 4 + B * (P * P * N + 512 * 1024) * 4
def e2p(): return equi2pers_patches(rgb, 80, 4, P, layout=lay)
def p2e(): return pers2equi(dp, 80, 4, P, (512, 1024), None, layout=lay)
for _ in range(5): e2p(); p2e()
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K): e2p(); p2e()
torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / K
print("one stream : %.1f us per pair  %.0f GB/s (%.1f %% of 8 TB/s)" % (t1 * 1e6, nbytes / t1 / 1e9, nbytes / t1 / 8e12 * 100))
sa, sb = _concurrent_streams(2, "cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        with torch.cuda.stream(sa): e2p()
        with torch.cuda.stream(sb): p2e()
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / K
    print("two streams: %.1f us per pair  %.0f GB/s (%.1f %% of 8 TB/s)" % (t2 * 1e6, nbytes / t2 / 1e9, nbytes / t2 / 8e12 * 100))
for ns in (2, 3, 4):
    ss = _concurrent_streams(ns, "cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        with torch.cuda.stream(ss[k % ns]): e2p(); p2e()
    torch.cuda.synchronize(); t3 = (time.perf_counter() - t0) / K
    print("%d streams, a whole pair each (round robin): %.1f us per pair  %.0f GB/s (%.1f %% of 8 TB/s)" % (ns, t3 * 1e6, nbytes / t3 / 1e9, nbytes / t3 / 8e12 * 100))
