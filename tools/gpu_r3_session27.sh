#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for V in 0 -1; do
echo "== LIDIFF_SIDE_PRIORITY=$V"
for i in 1 2 3; do LIDIFF_SIDE_PRIORITY=$V timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>&1 | grep -v amdgpu | tail -1 | cut -c40-75,190-230; done
done
LIDIFF_SIDE_PRIORITY=-1 timeout 300 python tools/debug/step_timeline.py 2>&1 | grep -v amdgpu | grep -v "^  step [0-9]" | head -3
