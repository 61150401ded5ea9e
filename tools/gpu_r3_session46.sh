#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_network.py -m gpu -q -x -k "bit_reproducible or gloo" 2>&1 | grep -E "^E  .*Assert|passed|failed|Error" | cut -c1-200
