#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s40
{
echo "# 256-row tiles (flags 0) vs 128-row tiles (flags 16 = LIDIFF_CONV_TILE_128) on the bench maps, CFG pair stacked: python tools/conv_probe.py --replicas 2 --cases ..."
for SG in 1.0 0.3; do
timeout 300 python tools/conv_probe.py --sigma $SG --replicas 2 --iters 30 --cases "2,64,64,k3,-1,0;2,64,64,k3,-1,16;2,32,64,k3,-1,0;2,32,64,k3,-1,16;1,64,64,k3,-1,0;1,64,64,k3,-1,16;0,32,32,down,-1,0;0,32,32,down,-1,16;1,64,64,down,-1,0;1,64,64,down,-1,16;1,32,32,k3,-1,0;1,32,32,k3,-1,16" 2>&1 | grep -v amdgpu
done
} > gpurun_out/s40/tile256_probe.txt
cat gpurun_out/s40/tile256_probe.txt | sed -E "s#m_in=[0-9]* ##; s#pairs=[0-9]* ##"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>&1 | grep -v amdgpu | tail -1 | cut -c40-75,190-230; done
