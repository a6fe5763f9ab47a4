#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_rife.py -x -q -m gpu -k "426 or 417 or quad" 2>&1 | tail -4
(timeout 120 python tools/rife_arch_bench.py --split 2>&1 | grep -v amdgpu.ids
echo "--- VFI_STAGE_QUAD=0 (cell kernels)"
VFI_STAGE_QUAD=0 timeout 120 python tools/rife_arch_bench.py --split --arch=4.26 --arch=4.17 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/arch_bench_quad.txt
