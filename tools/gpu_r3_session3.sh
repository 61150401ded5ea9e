#!/bin/bash
# Round-3 GPU session 3: probe-build ablations of the tile kernel (stride-8 256->256, one replica): which part of a stage costs what
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s3
mkdir -p $O
for P in 0 16 1 2 3 4 19 20 23; do
  for F in 0 8; do
    echo "-- probe $P flags $F" >> $O/ablate.txt
    timeout 120 python tools/conv_probe.py --timeline --probe $P --replicas 1 --flags $F --level 3 --cin 256 --cout 256 --iters 10 2>&1 | grep -v amdgpu | cut -c1-250 >> $O/ablate.txt
  done
done
cat $O/ablate.txt
