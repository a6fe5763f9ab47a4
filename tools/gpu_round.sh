#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_rife.py -x -q -m gpu -k "beta" 2>&1 | tail -8
