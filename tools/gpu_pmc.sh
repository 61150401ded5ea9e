#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc_r02; mkdir -p $OUT
PASSES="sq1 sq2" bash tools/pmc_probe.sh r02_tile --level 3 --cin 256 --cout 256 --iters 5 2>&1 | grep "^sq" | tee $OUT/tile.txt
PASSES="sq1 sq2" bash tools/pmc_probe.sh r02_bf16p1 --kernel bf16 --planes 1 --level 3 --cin 256 --cout 256 --iters 5 2>&1 | grep "^sq" | tee $OUT/bf16p1.txt
PASSES="sq1 sq2" bash tools/pmc_probe.sh r02_bf16p2 --kernel bf16 --planes 2 --level 3 --cin 256 --cout 256 --iters 5 2>&1 | grep "^sq" | tee $OUT/bf16p2.txt
