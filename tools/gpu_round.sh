#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dist_nodes.py tests/test_gpu_rife.py -x -q -m gpu 2>&1 | tail -3
REPS=3 timeout 200 python tools/node_e2e.py 65 1 2>&1 | grep "node e2e" | tee gpurun_out/node_e2e_b1.log
