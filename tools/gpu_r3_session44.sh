#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "batch_norm_train" 2>&1 | grep -E "^>|^E  .*Assert|test_gpu_kernels.py:[0-9]+|passed|failed" | cut -c1-220 | head -12
