#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python tools/splat_bench.py > gpurun_out/r05_splat_bench_binned.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05_splat_bench_binned.txt | cut -c1-330
(time timeout 1200 python -m pytest tests/test_gpu_m2m_ops.py tests/test_gpu_fuzz.py -x -q -m gpu) > gpurun_out/r05d_tests.log 2>&1
tail -12 gpurun_out/r05d_tests.log
timeout 300 python tools/m2m_bench.py > gpurun_out/r05d_m2m_bench.txt 2>&1; sed -n 2,3p gpurun_out/r05d_m2m_bench.txt; grep "splat" gpurun_out/r05d_m2m_bench.txt
