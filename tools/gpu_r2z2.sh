#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/r2z2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o b -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events --no-alt > /dev/null 2> $OUT/prof.err
cd $R
DB=$(find $OUT/prof -name '*.db' | head -1); python tools/rocpd_stats.py $DB --top 60 > $OUT/kernel_stats.md 2>&1; grep "lidiff::\|TOTAL" $OUT/kernel_stats.md | grep -v spconv | head -30; rm -rf $OUT/prof
