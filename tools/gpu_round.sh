#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rife.py tests/test_gpu_m2m.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/gpu_tests.log
timeout 200 python tools/rife_arch_bench.py 2>&1 | grep "^RIFE" | tee gpurun_out/rife_arch_bench.log
