#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
(time timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_m2m.py tests/test_gpu_m2m_ops.py tests/test_gpu_ifunet.py tests/test_gpu_ifrnet.py tests/test_gpu_rife40.py -x -q -m gpu) > gpurun_out/r05c_tests.log 2>&1
tail -15 gpurun_out/r05c_tests.log
timeout 300 python tools/m2m_bench.py > gpurun_out/r05c_m2m_bench.txt 2>&1; head -12 gpurun_out/r05c_m2m_bench.txt
timeout 300 python tools/ifunet_bench.py > gpurun_out/r05c_ifunet_bench.txt 2>&1; tail -4 gpurun_out/r05c_ifunet_bench.txt
timeout 300 python tools/ifrnet_bench.py > gpurun_out/r05c_ifrnet_bench.txt 2>&1; grep "ms/frame" gpurun_out/r05c_ifrnet_bench.txt
timeout 300 python tools/rife40_bench.py > gpurun_out/r05c_rife40_bench.txt 2>&1; tail -5 gpurun_out/r05c_rife40_bench.txt
