#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s24
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "row_kernel or pair_list or centre_tail" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -q -x -k "not every_network_conv and not late_trajectory and not maps_bit_exact and not gloo and not training and not bf16" 2>&1 | tail -4
for V in 0 8; do
  echo "== LIDIFF_CONV_FLAGS=$V"
  for i in 1 2 3; do LIDIFF_CONV_FLAGS=$V timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>/dev/null | cut -c40-75,190-230; done
done
