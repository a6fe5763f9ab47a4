#!/bin/bash
# round 3, first GPU call: Winograd kernel correctness + speed, then the RIFE network on it
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== winograd op tests"; timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "winograd" 2>&1 | tail -15
echo "== wino_bench"; timeout 240 python tools/wino_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids\|hipcc"
echo "== rife tests"; timeout 600 python -m pytest tests/test_gpu_rife.py -q -m gpu -x 2>&1 | tail -15
echo "== bench"; timeout 300 python bench.py --steps 5 --warmup 2 --no-e2e --no-extras --no-cpu-baseline 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "== bench direct"; VFI_CONV_WINOGRAD=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-e2e --no-extras --no-cpu-baseline 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-400
echo "== bocchi"; timeout 600 python -m pytest tests/test_gpu_bocchi.py -q -m gpu -s 2>&1 | grep -v "Comfy\|Warning" | tail -25
} 2>&1 | tee gpurun_out/r03a.log | tail -120
