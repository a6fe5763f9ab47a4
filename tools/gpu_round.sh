#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rife.py -x -q -m gpu -k "40" 2>&1 | tail -25 | tee gpurun_out/gpu_tests.log
