#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for ab in 0 1 128 256 512 4; do echo "== ablate $ab"; VFI_WINO_ABLATE=$ab timeout 120 python tools/wino_bench.py "res_c64 x32" "2440" "128->128 @540" 2>&1 | grep "rife\|film"; done
} 2>&1 | tee gpurun_out/r03f.log | tail -80
