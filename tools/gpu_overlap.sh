#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/overlap; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -q -m gpu -k "overlapped or completion or cfg_pair or end_to_end or c1_one or t50" --tb=short 2>&1 | grep -v "^    \|^$" | tail -3 | tee $OUT/tests.txt
for i in 1 2; do
timeout 400 python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-coords-roofline --no-kernel-events > $OUT/bench_u.json 2> $OUT/bench_u.err
python -c "
import json; j=json.load(open('$OUT/bench_u.json')); print('uncond on side: value %.2f ms %.2f'%(j['value'], j['ms_per_step']))"
done
