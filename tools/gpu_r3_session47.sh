#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -q -x -k "not gloo and not bf16 and not train and not every_network_conv and not late_traj and not maps_bit" 2>&1 | tail -3
for V in 1 0; do
  echo "== LIDIFF_CONV_LINEAR=$V"
  for i in 1 2 3; do LIDIFF_CONV_LINEAR=$V timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>&1 | grep -v amdgpu | tail -1 | cut -c40-75,190-230; done
done
