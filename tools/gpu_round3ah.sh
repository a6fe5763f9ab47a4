#!/bin/bash
# round 3ah: two-wave Winograd kernel without its chunk barrier (timing only: results wrong)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for v in 32 33 34; do echo "== VFI_WINO16_ABL=$v"; VFI_WINO16_ABL=$v timeout 120 python tools/wino_bench.py "res_c64 x32" 2>&1 | grep "rife" | sed 's/.*2-wave/2-wave/'; done
} 2>&1 | tee gpurun_out/r03ah.log | tail -10
