timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "=== soak B=16 2000"; REPS=2000 timeout 900 python tools/lanes_trace.py 2>&1 | grep -v amdgpu.ids | grep -v "      rows" | tail -3
echo "=== soak B=8 3000"; B=8 REPS=3000 timeout 900 python tools/lanes_trace.py 2>&1 | grep -v amdgpu.ids | grep -v "      rows" | tail -3
timeout 300 python tools/conc3.py 2>&1 | grep -v amdgpu.ids | tail -7
