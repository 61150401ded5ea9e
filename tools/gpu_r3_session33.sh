#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 300 python tools/conv_probe.py --replicas 1 --iters 30 --cases "0,3,32,k3,1,0;0,3,32,k3,1,8" 2>&1 | grep -v amdgpu | cut -c1-175
timeout 300 python tools/conv_probe.py --replicas 1 --iters 30 --centre-tail --cases "0,3,32,k3,1,8" 2>&1 | grep -v amdgpu | cut -c1-175
timeout 300 python tools/conv_probe.py --sigma 0.2 --replicas 1 --iters 30 --cases "0,3,32,k3,1,0;0,3,32,k3,1,8" 2>&1 | grep -v amdgpu | cut -c1-175
