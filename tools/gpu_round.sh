#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dist_nodes.py -x -q -m gpu 2>&1 | grep -v "^Comfy\|Gloo\|amdgpu.ids\|socket.cpp" | head -80 | tee gpurun_out/gpu_tests.log
