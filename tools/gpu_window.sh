#!/bin/bash
# rocprofv3 kernel trace of the driver's bench command (no per-launch events): main-queue busy/idle, idle gaps of the chip, the kernel
# sequence around one step boundary, per-kernel statistics.   usage: bash tools/gpu_window.sh <tag> [env assignments...]
T=${1:-window}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/bench.py --steps 20 --warmup 5 --no-kernel-events --no-cpu-baseline --no-train --no-alt --no-closed-loop --no-coords-roofline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py $DB --top 45 > $O/rocprofv3_kernel_stats.md 2>&1
python tools/rocpd_gaps.py $DB --last-ms ${LAST_MS:-560} --top 15 > $O/idle_gaps.txt 2>&1
python tools/rocpd_main_queue.py $DB --last-ms ${LAST_MS:-560} > $O/main_queue.txt 2>&1
python tools/rocpd_window.py $DB --after cfg_dpm_step_kernel --nth 3 --ms 4.0 > $O/step_boundary_window.txt 2>&1
if [ -z "${KEEP_DB:-}" ]; then rm -rf $O/prof; fi
cut -c1-200 $O/bench_under_rocprof.json; head -12 $O/main_queue.txt
