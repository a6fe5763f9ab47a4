#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rife.py -q -m gpu -x -k dtype 2>&1 | tail -5
bash tools/profile_other_nodes_r05.sh r05 2>&1 | tail -40
