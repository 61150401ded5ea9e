#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 300 python tools/debug/step_timeline.py 2>&1 | grep -v amdgpu | grep -v "^  step [0-9]"
