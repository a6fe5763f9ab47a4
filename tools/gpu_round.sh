#!/bin/bash
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_ifrnet.py tests/test_gpu_m2m.py -x -q -m gpu -k "1080p_node_default or node or plan" 2>&1 | tail -4
