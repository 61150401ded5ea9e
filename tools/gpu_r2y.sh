#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2y; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -q -m gpu -k "bf16 or backward" --tb=short -s 2>&1 | grep -v "^    \|^$" | tail -25 > $OUT/tests.txt; grep "step:\|passed\|failed\|^E " $OUT/tests.txt | cut -c1-300
timeout 300 python tools/train_probe.py --steps 3 --precision bf16 2>&1 | tail -1 | tee $OUT/train_bf16.txt
