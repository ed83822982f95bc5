#!/bin/bash
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 120 python tools/kbench.py --layout ref --B 1
