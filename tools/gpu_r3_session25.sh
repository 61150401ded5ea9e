#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for V in 0 8; do
  echo "== LIDIFF_CONV_FLAGS=$V"
  for i in 1 2 3; do LIDIFF_CONV_FLAGS=$V timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>&1 | grep -v amdgpu | tail -1 | cut -c40-75,190-230; done
done
