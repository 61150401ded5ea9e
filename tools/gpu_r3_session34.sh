#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "row_kernel or pair_list" 2>&1 | tail -3
C=""
for s in "3,256,256" "2,256,128" "2,128,128" "1,128,96" "0,96,96"; do C="$C$s,up,0,0;$s,up,0,8;"; done
timeout 600 python tools/conv_probe.py --replicas 2 --iters 30 --up-ordered --cases "${C%;}" 2>&1 | grep -v amdgpu | cut -c1-175
