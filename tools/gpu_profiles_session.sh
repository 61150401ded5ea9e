#!/bin/bash
# GPU profiling session: PMC traffic passes, rocprofv3 kernel stats of the bench command, CPU-baseline thread probe.
set -u
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/profiles; mkdir -p $OUT
bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; tail -5 $OUT/pmc.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt --no-coords-roofline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
cd $R
DB=$(find $OUT/prof -name '*.db' | head -1); python tools/rocpd_stats.py $DB --top 45 > $OUT/kernel_stats.md 2>&1; head -12 $OUT/kernel_stats.md; rm -rf $OUT/prof
timeout 400 python tools/cpu_threads_probe.py 32 64 2>&1 | grep threads | tee $OUT/cpu_threads.txt
