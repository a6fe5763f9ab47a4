#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_m2m.py tests/test_gpu_rife.py tests/test_gpu_film.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/gpu_tests.log
for v in "" 43 44 12; do
echo "VFI_GROUPED_VARIANT=$v" | tee -a gpurun_out/m2m_variants.log
VFI_GROUPED_VARIANT=$v timeout 200 python tools/m2m_bench.py 2>&1 | grep -E "prepare|deconv|pool_mean" | tee -a gpurun_out/m2m_variants.log
done
