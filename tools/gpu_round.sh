#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_ifrnet.py -q -m gpu 2>&1 | tail -150 > gpurun_out/ifrnet_tests.log
tail -5 gpurun_out/ifrnet_tests.log
