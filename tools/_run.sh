timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/kbench_bwd.py 2>&1 | grep -v amdgpu.ids | grep "planar\|reference"
