#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python tools/wino_ledger.py --var=1 --var=2 > gpurun_out/r05_wino_variants.txt 2>&1
tail -32 gpurun_out/r05_wino_variants.txt
