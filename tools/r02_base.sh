#!/bin/bash
# baseline resample numbers at every BASELINE configuration (run on the GPU box from the repo root)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
{
python tools/kbench.py --B 8 --P 256 --copy
python tools/kbench.py --B 8 --P 128
python tools/kbench.py --B 1 --P 256
python tools/kbench.py --B 16 --P 256
python tools/kbench.py --B 1 --P 256 --H 1024 --W 2048 --nrows 6
python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6
python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6 --half
python tools/kbench.py --B 4 --P 512 --H 2048 --W 4096 --nrows 6 --half
} 2>&1 | tee $O/r02_base.log
