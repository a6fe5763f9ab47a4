#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_m2m_ops.py tests/test_gpu_m2m.py -x -q -m gpu 2>&1 | tail -3
timeout 200 python tools/splat_bench.py 2>&1 | grep "softsplat" | tee gpurun_out/splat_bench_v4.log
timeout 200 python tools/m2m_bench.py 2>&1 | grep -E "prepare|softsplat" | tee -a gpurun_out/splat_bench_v4.log
