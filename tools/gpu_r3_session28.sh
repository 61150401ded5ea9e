#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "single_read or voxel_hash or centre_tail" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -q -x -k "not every_network_conv and not late_trajectory and not gloo and not training and not bf16" 2>&1 | tail -4
for V in 1 0; do
  echo "== LIDIFF_SINGLE_READ=$V"
  for i in 1 2 3; do LIDIFF_SINGLE_READ=$V timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>&1 | grep -v amdgpu | tail -1 | cut -c40-75,190-230; done
done
timeout 300 python tools/debug/step_timeline.py 2>&1 | grep -v amdgpu | grep -v "^  step [0-9]" | head -14
