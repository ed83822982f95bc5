#!/bin/bash
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4
FLAG=fold_point_feat timeout 900 python tools/flag_ab.py
