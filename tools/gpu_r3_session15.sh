#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s15
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -q -k "not every_network_conv and not late_trajectory and not maps_bit_exact and not gloo and not training and not bf16" 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>/dev/null | cut -c1-230; done
timeout 300 python tools/debug/step_timeline.py 2>&1 | grep -v amdgpu | head -12
