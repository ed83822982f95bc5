#!/bin/bash
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python - <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from omnifusion_amd.model.spherical_model_iterative import spherical_fusion as sf_it
from omnifusion_amd.model._engine import Engine
from omnifusion_amd.weights import make_state_dict
net3 = sf_it(6, 46, (128, 128), (80, 80)).cuda(); net3.load_state_dict(make_state_dict(42, 46, True))
x3 = torch.rand(1, 3, 1024, 2048, device="cuda")
def bench(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for rnd in range(3):
    for rg in (True, False):
        Engine.rows_gemm = rg
        print("cfg3 iterative iter=2 B=1 1024x2048 nrows=6, rows_gemm %s: %.3f ms" % (rg, bench(lambda: net3(x3, 2)) * 1e3), flush=True)
PY
