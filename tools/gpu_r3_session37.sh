#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 300 python tools/debug/boundary_probe.py 2>&1 | grep -v amdgpu | tail -11
