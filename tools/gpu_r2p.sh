#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2p; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_kernels.py -x -q -m gpu -k "training_step or backward_vs_oracle or two_rank or refine" -s 2>&1 | tail -12 | tee $OUT/tests.txt
