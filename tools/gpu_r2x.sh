#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2x; mkdir -p $OUT
timeout 180 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fps or farthest" --tb=short -s 2>&1 | grep -v "^    \|^$" | tail -12 | tee $OUT/tests.txt
