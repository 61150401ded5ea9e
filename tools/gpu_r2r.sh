#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2r; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16" --tb=short 2>&1 | grep -v "^    \|^$" | tail -40 > $OUT/tests.txt; tail -3 $OUT/tests.txt
for pl in 1 2 3; do
  for shape in "3 256 256" "3 384 256" "4 256 256" "3 128 128" "3 128 256" "2 128 128" "2 64 128"; do
    set -- $shape
    timeout 120 python tools/conv_probe.py --kernel bf16 --planes $pl --level $1 --cin $2 --cout $3 --iters 10 2>&1 | grep sigma | sed "s/^/planes=$pl /" | tee -a $OUT/sweep.txt
  done
done
