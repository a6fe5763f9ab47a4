#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== clip run + m2m ops"; timeout 900 python -m pytest tests/test_gpu_clip_run.py tests/test_gpu_m2m_ops.py tests/test_gpu_m2m.py -q -m gpu 2>&1 | tail -12
echo "== m2m bench"; timeout 200 python tools/m2m_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | head -12
echo "== 2-rank bench plumbing (gloo, one GPU)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --batch 4 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 | cut -c1-1500
} 2>&1 | tee gpurun_out/r03k.log | tail -60
