#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -q -x -k "not every_network_conv and not late_trajectory and not maps_bit_exact and not gloo and not training and not bf16" 2>&1 | tail -12
for V in "LIDIFF_LAZY_XT=1" "LIDIFF_LAZY_XT=0"; do
  echo "== $V"
  for i in 1 2 3; do env $V timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>/dev/null | cut -c40-75,190-230; done
done
timeout 300 python tools/debug/step_timeline.py 2>&1 | grep -v amdgpu | head -14
