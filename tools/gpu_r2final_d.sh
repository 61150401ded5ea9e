#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2fd; mkdir -p $OUT
timeout 600 python bench.py --steps 50 --warmup 2 --cached-condition --no-cpu-baseline > $OUT/bench_steps50.json 2> $OUT/bench_steps50.err
python -c "
import json; j=json.load(open('$OUT/bench_steps50.json')); print('value %.2f ms %.2f cached %.2f alt %.2f frac %.3f'%(j['value'], j['ms_per_step'], j['cached_condition']['value'], j['alt']['value'], j['roofline']['frac']))"
