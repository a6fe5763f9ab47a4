#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== winograd op tests"; timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "winograd" 2>&1 | tail -3
echo "== wino_bench"; timeout 240 python tools/wino_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids\|hipcc"
for ab in 1; do echo "== ablate $ab"; VFI_WINO_ABLATE=$ab timeout 120 python tools/wino_bench.py "res_c64 x32" "2440" 2>&1 | grep "rife\|film"; done
} 2>&1 | tee gpurun_out/r03g.log | tail -80
