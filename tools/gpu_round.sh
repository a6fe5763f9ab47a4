#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rife.py -x -q -m gpu -k "426" 2>&1 | tail -12 | tee gpurun_out/gpu_tests.log
timeout 200 python tools/rife_arch_bench.py --split 2>&1 | grep -A1 "^RIFE 4.26" | tee gpurun_out/rife_arch_bench2.log
