python tools/e2p_check.py 2>&1 | grep -v "amdgpu.ids\|omni\]" | head -12
python tools/kbench.py --B 8 --P 256 2>&1 | grep -v amdgpu
python tools/kbench.py --B 8 --P 128 2>&1 | grep -v amdgpu
python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6 --half 2>&1 | grep -v amdgpu
