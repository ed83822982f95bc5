timeout 300 python tools/pair_conc.py 2>&1 | grep -v amdgpu.ids
