#!/bin/bash
# one GPU-box visit: M2M tests, full gpu suite, bench.  Outputs under gpurun_out/.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_gpu_m2m.py tests/test_gpu_m2m_ops.py -x -q -m gpu > gpurun_out/m2m_tests.log 2>&1
echo "m2m rc=$?" >> gpurun_out/m2m_tests.log
tail -30 gpurun_out/m2m_tests.log
timeout 420 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rife.py tests/test_gpu_film.py -x -q -m gpu > gpurun_out/rest_tests.log 2>&1
echo "rest rc=$?" >> gpurun_out/rest_tests.log
tail -5 gpurun_out/rest_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
tail -2 gpurun_out/bench.log
