#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s26
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --layer-table gpurun_out/s26/layer_table.txt > gpurun_out/s26/bench.json 2>/dev/null
cat gpurun_out/s26/layer_table.txt | cut -c1-150
python - <<'PY'
import json
d = json.load(open("gpurun_out/s26/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("roofline_narrow", {}).get("frac"), d.get("roofline_narrow", {}).get("ms_per_step"))
PY
