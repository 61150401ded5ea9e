#!/bin/bash
# row kernel (spconv_rows.hip): parity, per-layer A/B against the tile kernel (flags 8 = LIDIFF_CONV_TILE_ONLY), bench A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s21
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "row_kernel or centre_tail or dense_kernel" 2>&1 | tail -5
C=""
for s in "0,96,96" "0,128,96" "0,32,32" "1,32,32" "1,96,96" "1,128,96" "1,32,64" "2,64,128" "2,192,128" "2,64,64" "3,128,256" "3,128,128" "4,128,256"; do
  C="$C$s,k1,0,0;$s,k1,0,8;"
done
timeout 600 python tools/conv_probe.py --replicas 2 --iters 30 --cases "${C%;}" 2>&1 | grep -v amdgpu | tee gpurun_out/s21/probe_k1.txt | cut -c1-170
for V in 0 8; do
  echo "== LIDIFF_CONV_FLAGS=$V"
  for i in 1 2 3; do LIDIFF_CONV_FLAGS=$V timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>/dev/null | cut -c40-75,190-230; done
done
