#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 300 python tools/gmfss_bench.py --coherent > gpurun_out/r05_gmfss_bench_merged.txt 2>&1; grep "rep \|sum" gpurun_out/r05_gmfss_bench_merged.txt | cut -c1-400
(time timeout 3000 python -m pytest tests -q -m gpu -x) > gpurun_out/r05_full_gpu_suite.log 2>&1
tail -12 gpurun_out/r05_full_gpu_suite.log
