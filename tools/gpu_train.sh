#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/train; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_kernels.py -q -m gpu -k "training_step or training_steps or bf16" --tb=short -s 2>&1 | grep -v "^    \|^$" | tail -12 | tee $OUT/tests.txt
timeout 300 python tools/train_probe.py --steps 3 --precision bf16 2>&1 | tail -1 | tee $OUT/train_bf16.txt
