#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== pytest node tests"; timeout 600 python -m pytest tests/test_gpu_rife.py -q -m gpu --no-header -rf 2>&1 | tail -8
echo "=== node e2e"; timeout 300 python tools/node_e2e.py 33 8 2>&1 | grep "node e2e"
timeout 300 python tools/node_e2e.py 33 1 2>&1 | grep "node e2e"
