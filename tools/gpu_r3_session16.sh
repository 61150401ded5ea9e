#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 300 python tools/debug/chain_probe.py --sigma 1.0 2>&1 | grep -v amdgpu
timeout 300 python tools/debug/chain_probe.py --sigma 0.3 2>&1 | grep -v amdgpu
