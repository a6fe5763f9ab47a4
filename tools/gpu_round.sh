#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 700 python -m pytest tests/test_gpu_rife.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/gpu_tests.log
