#!/bin/bash
# rocprofv3 kernel statistics of the training step (tools/train_probe.py), fp32 and bf16
set -u
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/trainprof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pr in bf16 32; do
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$pr -o t -- python $R/tools/train_probe.py --steps 3 --precision $pr > $OUT/train_$pr.log 2> $OUT/prof_$pr.err
DB=$(find $OUT/prof_$pr -name '*.db' | head -1); python $R/tools/rocpd_stats.py $DB --top 30 > $OUT/kernel_stats_$pr.md 2>&1; rm -rf $OUT/prof_$pr
tail -1 $OUT/train_$pr.log; head -6 $OUT/kernel_stats_$pr.md
done
