#!/bin/bash
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_gpu_rife.py -x -q -m gpu -k "dtype_widget or rejects or node_against_reference_golden" 2>&1 | tail -4
