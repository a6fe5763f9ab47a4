#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python tools/option_ab.py wino_quant 0 > gpurun_out/r05_wino_quant_ab.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05_wino_quant_ab.txt
