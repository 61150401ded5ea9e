#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/r2v; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o train -- python $R/tools/train_probe.py --steps 3 --precision bf16 > $OUT/train.log 2> $OUT/prof.err
cd $R
DB=$(find $OUT/prof -name '*.db' | head -1); python tools/rocpd_stats.py $DB --top 40 > $OUT/kernel_stats.md 2>&1; head -42 $OUT/kernel_stats.md; rm -rf $OUT/prof
