#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== pytest film"; timeout 900 python -m pytest tests/test_gpu_film.py -q -m gpu --no-header -rf 2>&1 | tail -40 | cut -c1-400
