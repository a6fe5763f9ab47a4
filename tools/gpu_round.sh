#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rife.py tests/test_gpu_m2m.py -x -q -m gpu -k "large" 2>&1 | tail -12
