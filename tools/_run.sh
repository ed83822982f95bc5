OPT=conv_nohalo VALS=0,1 timeout 300 python tools/pipe_ab.py 2>&1 | grep -v amdgpu.ids
