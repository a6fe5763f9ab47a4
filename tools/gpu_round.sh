#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/thp.log
for thp in 1 0 1 0; do
echo "VFI_HOST_THP=$thp" | tee -a gpurun_out/thp.log
VFI_HOST_THP=$thp REPS=3 timeout 200 python tools/node_e2e.py 65 8 2>&1 | grep "node e2e" | cut -c1-110 | tee -a gpurun_out/thp.log
done
