#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "thin_input or centre_tail or spconv_fwd or stem" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -q -x -k "not every_network_conv and not late_trajectory and not maps_bit_exact and not gloo and not bf16" 2>&1 | tail -4
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>&1 | grep -v amdgpu | tail -1 | cut -c40-75,190-230; done
