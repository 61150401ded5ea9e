#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/bf16tile; mkdir -p $OUT; rm -f $OUT/sweep.txt
LIDIFF_BF16_TILE=64 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16" --tb=short 2>&1 | tail -2 | tee $OUT/tests.txt
for cfg in "128 64" "64 64" "64 32"; do
  set -- $cfg
  for shape in "3 256 256" "3 384 256" "4 256 256" "3 128 128" "2 128 128"; do
    set -- $cfg $shape
    LIDIFF_BF16_TILE=$1 LIDIFF_BF16_KS=$2 timeout 120 python tools/conv_probe.py --kernel bf16 --planes 1 --level $3 --cin $4 --cout $5 --iters 10 2>&1 | grep sigma | sed "s/^/tile=$1 ks=$2 /" | awk '{print $1,$2,$4,$6,$13,$14}' | tee -a $OUT/sweep.txt
  done
done
