#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_gmfss.py -q -m gpu -s 2>&1 | grep -v "Warning\|_VF\|amdgpu.ids" > gpurun_out/gmfss_tests.log
grep "^GMFSS\|passed\|failed" gpurun_out/gmfss_tests.log | cut -c1-500
