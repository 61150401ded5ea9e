#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "256_row" 2>&1 | tail -3
for SG in 1.0 0.3; do
timeout 300 python tools/conv_probe.py --sigma $SG --replicas 2 --iters 30 --cases "2,64,64,k3,-1,0;2,64,64,k3,-1,16;2,32,64,k3,-1,0;2,32,64,k3,-1,16;1,64,64,k3,-1,0;1,64,64,k3,-1,16" 2>&1 | grep -v amdgpu | sed -E "s#m_in=[0-9]* ##; s#pairs=[0-9]* ##"
done
for V in 0 16; do
  echo "== LIDIFF_CONV_FLAGS=$V"
  for i in 1 2 3; do LIDIFF_CONV_FLAGS=$V timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>&1 | grep -v amdgpu | tail -1 | cut -c40-75,190-230; done
done
