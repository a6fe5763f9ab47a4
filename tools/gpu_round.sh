#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python tools/rife40_bench.py 2>&1 | grep -A1 "^RIFE 4.0" | tee gpurun_out/rife40_bench.log
REPS=2 timeout 200 python tools/node_e2e.py 17 4 2>&1 | grep "node e2e" | tee gpurun_out/node_e2e_misc.log
