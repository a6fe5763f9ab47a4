#!/bin/bash
# round 3ae: two-wave Winograd kernel with groups by SIMD id and de-interleaved patch columns
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== parity"; timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "two_wave" 2>&1 | tail -3
echo "== wino_bench"; timeout 240 python tools/wino_bench.py rife 2>&1 | grep "rife" | sed 's/ |.*| 2-wave/ | 2-wave/'
for v in 1 2; do echo "== ABL=$v"; VFI_WINO16_ABL=$v timeout 120 python tools/wino_bench.py "res_c64 x32" 2>&1 | grep "rife" | sed 's/.*2-wave/2-wave/'; done
} 2>&1 | tee gpurun_out/r03ae.log | tail -30
