#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for SG in 1.0 0.2; do
C=""
for s in "0,96,96" "0,128,96" "0,32,32" "1,32,32" "1,96,96"; do C="$C$s,k3,1,0;$s,k3,1,8;"; done
timeout 600 python tools/conv_probe.py --sigma $SG --replicas 2 --iters 30 --centre-tail --cases "${C%;}" 2>&1 | grep -v amdgpu | cut -c1-175
done
