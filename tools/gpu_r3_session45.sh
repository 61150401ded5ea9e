#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1200 python -m pytest tests/test_gpu_network.py tests/test_gpu_bf16_blocks.py -m gpu -q -x -k "train or bf16 or gloo or grad or step" 2>&1 | tail -3
for V in 1 0; do
echo "== LIDIFF_MATCHES_AHEAD=$V"
LIDIFF_MATCHES_AHEAD=$V timeout 300 python tools/train_probe.py --steps 4 --precision bf16 2>&1 | grep -v amdgpu | tail -1
done
