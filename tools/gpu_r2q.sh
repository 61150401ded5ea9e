#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2q; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -q -m gpu -k "training_step and bf16" -s --tb=short 2>&1 | grep -v "^    \|^$" | tail -150 > $OUT/tests.txt
