timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -3
