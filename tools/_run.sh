timeout 900 python -m pytest tests/test_eval_gpu.py tests/test_io_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/eval.py --batches 12 --batch 8 --ply-every 6 --out /tmp/res 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python tools/eval.py --batches 12 --batch 8 --depth 1 --out /tmp/res 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/eval.py --batches 6 --batch 4 --iterative --out /tmp/res 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py --steps 50 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}); print(d['host_fed']['panoramas_per_s_per_gpu'], d['host_fed']['frac_of_resident'])"
