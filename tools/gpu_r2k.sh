#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2k; mkdir -p $OUT
P="timeout 120 python tools/conv_probe.py --iters 10 --probe 0 --timeline"
for shape in "0 96 96 k3" "0 128 96 k3" "1 96 96 k3" "0 32 32 k3" "1 32 64 k3" "2 64 64 k3" "0 96 96 k1" "2 128 128 k3"; do
  set -- $shape
  $P --level $1 --cin $2 --cout $3 --kind $4 2>&1 | grep -v "amdgpu.ids\|shader clock" | tee -a $OUT/lowdens.txt
done
