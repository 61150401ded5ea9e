#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s5
mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events --cached-condition > $O/bench_cached.json 2> $O/bench_cached.err
python - <<'PY' 2>&1 | tee $O/summary.txt
import json
d=json.load(open("gpurun_out/s5/bench_cached.json"))
print("value", d["value"], "ms", d["ms_per_step"], "cached", d["cached_condition"]["value"], d["cached_condition"]["ms_per_step"])
PY
