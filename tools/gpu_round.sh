#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_film.py tests/test_gpu_m2m.py tests/test_gpu_dist_nodes.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/node_e2e_models.py 2>&1 | grep "node e2e" | tee gpurun_out/node_e2e_models.log
