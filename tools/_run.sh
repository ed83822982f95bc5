timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
echo "=== soak B=16 1500"; REPS=1500 timeout 900 python tools/lanes_trace.py 2>&1 | grep -v amdgpu.ids | grep -v "      rows" | tail -2
timeout 300 python tools/plain_ab.py 2>&1 | grep -v amdgpu.ids | grep "tile -1\|tile  6"
