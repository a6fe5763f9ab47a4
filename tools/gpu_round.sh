#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
(timeout 20 python -m pytest tests/test_gpu_gmfss.py -q -m gpu -s -k "64-64" 2>&1 | grep "^GMFSS\|passed\|failed\|Error" | cut -c1-420
timeout 20 python tools/gmfss_bench.py 2>&1 | grep -v "Warning\|_VF\|amdgpu.ids\|hipcc") | tee gpurun_out/gmfss_bench_v2.txt | tail -7
