#!/bin/bash
# round 3ag: two-wave Winograd kernel, group 0's epilogue moved behind the next barrier
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== parity"; timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "two_wave" 2>&1 | tail -2
echo "== wino_bench"; timeout 240 python tools/wino_bench.py rife 2>&1 | grep "rife" | sed 's/ |.*| 2-wave/ | 2-wave/'
} 2>&1 | tee gpurun_out/r03ag.log | tail -12
