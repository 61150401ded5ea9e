#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/s29
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt --no-coords-roofline --no-train > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_window.py $DB --nth 3 --ms 4.5 > $O/window.txt 2>&1
python tools/rocpd_gaps.py $DB --min-us 10 --top 8 --last-ms 300 | head -3
cat $O/window.txt
rm -rf $O/prof
