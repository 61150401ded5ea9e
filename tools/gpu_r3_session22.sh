#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s22
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "row_kernel" 2>&1 | tail -3
C=""
for s in "0,96,96" "0,128,96" "0,32,32" "1,32,64" "2,64,128" "2,192,128" "3,128,256"; do C="$C$s,k1,0,0;"; done
timeout 600 python tools/conv_probe.py --replicas 2 --iters 30 --cases "${C%;}" 2>&1 | grep -v amdgpu | cut -c1-170
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>/dev/null | cut -c40-75,190-230; done
timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --layer-table gpurun_out/s22/layer_table.txt > gpurun_out/s22/bench.json 2>/dev/null
cat gpurun_out/s22/layer_table.txt
