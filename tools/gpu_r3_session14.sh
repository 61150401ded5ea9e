#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s14
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "spconv or stem" 2>&1 | tail -2
D="3,256,256,k3,-1,0;3,384,256,k3,-1,0;3,128,128,k3,-1,0;3,128,256,k3,-1,0;3,64,128,k3,-1,0;4,256,256,k3,-1,0;4,128,256,k3,-1,0;3,384,256,k1,-1,0;2,128,128,k3,0,0;2,192,128,k3,0,0"
timeout 300 python tools/conv_probe.py --replicas 2 --sigma 1.0 --cases "$D" 2>&1 | grep sigma | cut -c1-200 | tee $O/probe.txt
for C in "3 256 256" "3 128 128"; do set -- $C
  timeout 200 python tools/conv_probe.py --timeline --replicas 1 --sparse-hint 0 --level $1 --cin $2 --cout $3 2>&1 | grep -v amdgpu | cut -c1-250 | tee -a $O/timelines.txt
done
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>/dev/null | cut -c1-230; done
