#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python tools/deconv_ab.py > gpurun_out/r05_deconv_ab.txt 2>&1; cat gpurun_out/r05_deconv_ab.txt | grep -v amdgpu.ids
(time timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_m2m.py tests/test_gpu_m2m_ops.py tests/test_gpu_ifunet.py tests/test_gpu_ifrnet.py -x -q -m gpu) > gpurun_out/r05c_tests.log 2>&1
tail -15 gpurun_out/r05c_tests.log
