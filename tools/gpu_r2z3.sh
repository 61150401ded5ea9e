#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/r2z3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline.py tests/test_gpu_network.py -q -m gpu -k "hash or maps or golden or unet or completion" --tb=short 2>&1 | tail -2 | tee $OUT/tests.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o b -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events --no-alt > $OUT/bench.json 2> $OUT/prof.err
cd $R
DB=$(find $OUT/prof -name '*.db' | head -1); python tools/rocpd_stats.py $DB --top 60 > $OUT/kernel_stats.md 2>&1; grep "kernel_map\|Memset\|fillBuffer\|TOTAL" $OUT/kernel_stats.md; rm -rf $OUT/prof
python -c "
import json; j=json.load(open('gpurun_out/r2z3/bench.json')); print(j['value'], j['roofline_hbm']['ms'], j['roofline_hbm']['frac'])"
