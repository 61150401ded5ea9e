#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for V in "" "HSA_ENABLE_INTERRUPT=0"; do
  echo "== env: $V"
  for i in 1 2 3; do env $V timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>/dev/null | cut -c40-75,190-230; done
  env $V timeout 300 python tools/debug/step_timeline.py 2>&1 | grep -v amdgpu | head -1
  env $V timeout 300 python tools/debug/chain_probe.py 2>&1 | grep -v amdgpu
done
