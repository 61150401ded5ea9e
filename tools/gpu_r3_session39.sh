#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "256_row" 2>&1 | grep -E "^E  |tests/test_gpu_kernels.py:[0-9]+|passed|failed" | head -12
