#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s7
mkdir -p $O
timeout 300 python tools/debug/step_timeline.py > $O/step_timeline.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_baseline.py tests/test_gpu_network.py -m gpu -q -k "not c1 and not t50 and not gloo" > $O/pytest.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --layer-table $O/layers.txt > $O/bench.json 2> $O/bench.err
cat $O/step_timeline.txt | grep -v amdgpu; tail -3 $O/pytest.txt; cut -c1-250 $O/bench.json; head -12 $O/layers.txt
