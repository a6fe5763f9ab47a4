#!/bin/bash
# round 3ac: first run of the two-waves-per-SIMD Winograd kernel (conv_wino16_kernel)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== parity"; timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "two_wave" 2>&1 | tail -15
echo "== wino_bench"; timeout 240 python tools/wino_bench.py rife "64->64" 2>&1 | grep "rife\|film"
} 2>&1 | tee gpurun_out/r03ac.log | tail -60
