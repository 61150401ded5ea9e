#!/bin/bash
# rocprofv3 kernel trace of the driver's bench command -> the kernel sequence around one step boundary (tools/rocpd_window.py)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/window
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt --no-coords-roofline --no-train > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_window.py $DB --nth 3 --ms 5.0 > $O/window.txt 2>&1
rm -rf $O/prof
grep -c . $O/window.txt
