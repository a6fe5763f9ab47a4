#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 38 python tools/gmfss_bench.py 2>&1 | grep -v "Warning\|_VF\|amdgpu.ids\|hipcc" | tee gpurun_out/gmfss_bench.txt | tail -6
