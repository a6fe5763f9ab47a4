#!/bin/bash
# round 3ad: where the two-wave Winograd kernel's time goes (runtime ablations; results wrong by design)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for v in 0 1 2 4 8 16 3; do
echo "== VFI_WINO16_ABL=$v"; VFI_WINO16_ABL=$v timeout 120 python tools/wino_bench.py "res_c64 x32" 2>&1 | grep "rife" | sed 's/.*2-wave/2-wave/'
done
} 2>&1 | tee gpurun_out/r03ad.log | tail -30
