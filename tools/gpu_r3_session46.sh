#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do timeout 900 python -m pytest tests/test_gpu_network.py -m gpu -q -x -s -k "gloo" 2>&1 | grep -E "a18|passed|failed" | cut -c1-160; done
