#!/bin/bash
# all 50 trajectory positions of the T = 50 schedule (the default bench samples 10 of them), with the conditions-encoded-once variant beside it
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/steps50
timeout 900 python bench.py --steps 50 --warmup 2 --cached-condition --no-cpu-baseline --no-train > gpurun_out/steps50/bench.json 2> gpurun_out/steps50/bench.err
python -c "
import json
d=json.load(open('gpurun_out/steps50/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['serial']['frac'], d.get('cached_condition',{}).get('value'), d.get('alt',{}).get('value'))"
