#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -q -x -k "not every_network_conv and not late_trajectory and not maps_bit_exact and not gloo and not bf16" 2>&1 | tail -3
for i in 1 2 3 4; do timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>&1 | grep -v amdgpu | tail -1 | cut -c40-75,190-230; done
timeout 300 python tools/debug/step_timeline.py 2>&1 | grep -v amdgpu | grep -E "^wall"
timeout 300 python tools/debug/boundary_probe.py 2>&1 | grep -v amdgpu | tail -8
