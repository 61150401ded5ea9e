#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
PROBE_USER_STREAM=1 timeout 300 python tools/debug/boundary_probe.py 2>&1 | grep -v amdgpu | tail -8
