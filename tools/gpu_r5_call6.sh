#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python tools/deconv_ab.py > gpurun_out/r05_deconv_ab.txt 2>&1; grep -v "amdgpu.ids\|     deconv" gpurun_out/r05_deconv_ab.txt
timeout 300 python tools/gmfss_bench.py --coherent > gpurun_out/r05_gmfss_bench.txt 2>&1; grep "rep " gpurun_out/r05_gmfss_bench.txt
