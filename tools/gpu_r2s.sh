#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2s; mkdir -p $OUT
for pl in 1 2; do
  for pr in 0 1 2 3 4 8 17 32 35 44; do
    LIDIFF_BF16_PROBE=$pr timeout 120 python tools/conv_probe.py --kernel bf16 --planes $pl --level 3 --cin 256 --cout 256 --iters 10 2>&1 | grep sigma | sed "s/^/planes=$pl probe=$pr /" | awk '{print $1,$2,$13,$14}' | tee -a $OUT/probe.txt
  done
done
