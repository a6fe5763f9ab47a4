#!/bin/bash
# round 3aa: FILM flow-estimator input as a channel window (no feature copies)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_film.py tests/test_gpu_bocchi.py tests/test_gpu_clip_run.py -m gpu -x -q 2>&1 | tail -3
echo "== film bench"; timeout 300 python tools/film_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | head -12
} 2>&1 | tee gpurun_out/r03aa.log | tail -40
