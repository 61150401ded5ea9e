#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for RB in 1 2; do
echo "== LIDIFF_ROWS_RB=$RB"
export LIDIFF_ROWS_RB=$RB

C=""
for s in "0,96,96" "0,128,96" "0,32,32" "1,32,64" "2,64,128" "2,192,128" "3,128,256" "2,64,64" "3,128,128"; do C="$C$s,k1,0,0;"; done
timeout 600 python tools/conv_probe.py --replicas 2 --iters 30 --cases "${C%;}" 2>&1 | grep -v amdgpu | sed -E "s#sigma=.* kind=##; s#m_in=.*reps=2 flags=0##"
C=""
for s in "2,128,128" "1,128,96" "0,96,96"; do C="$C$s,up,0,0;"; done
timeout 600 python tools/conv_probe.py --replicas 2 --iters 30 --up-ordered --cases "${C%;}" 2>&1 | grep -v amdgpu | sed -E "s#sigma=.* kind=##; s#m_in=.*reps=2 flags=0##"
done
