timeout 900 python -m pytest tests/test_io_gpu.py -x -q -m gpu 2>&1 | tail -3
