#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in 0 1 2; do echo "=== splat mode $m"; VFI_SPLAT_MODE=$m timeout 300 python tools/splat_bench.py 2>&1 | grep -E "softsplat" | cut -c1-250; done
echo "=== tests mode 1"; VFI_SPLAT_MODE=1 timeout 300 python -m pytest tests/test_gpu_m2m_ops.py -q -m gpu --no-header -k splat 2>&1 | tail -3
