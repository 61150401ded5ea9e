#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2t; mkdir -p $OUT
for tile in 64 128; do
LIDIFF_BF16_TILE=$tile timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16" --tb=short 2>&1 | grep -v "^    \|^$" | tail -40 > $OUT/tests_$tile.txt; tail -1 $OUT/tests_$tile.txt
done
for pl in 1 2; do
  for shape in "3 256 256" "3 384 256" "4 256 256" "3 128 128" "3 128 256" "2 128 128" "2 64 128"; do
    set -- $shape
    LIDIFF_BF16_TILE=64 timeout 120 python tools/conv_probe.py --kernel bf16 --planes $pl --level $1 --cin $2 --cout $3 --iters 10 2>&1 | grep sigma | sed "s/^/tile=64 planes=$pl /" | tee -a $OUT/sweep.txt
  done
done
