#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== m2m bench"; VFI_TRACE_SHAPES=1 timeout 200 python tools/m2m_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | head -50
echo "== m2m bench direct"; VFI_CONV_WINOGRAD=0 VFI_TRACE_SHAPES=1 timeout 200 python tools/m2m_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | head -30
} 2>&1 | tee gpurun_out/r03m.log | tail -90
