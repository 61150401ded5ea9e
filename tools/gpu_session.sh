# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4n}
O=$PWD/gpurun_out/$T; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-alt --no-coords-roofline --no-train --no-closed-loop > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_main_queue.py $DB --last-ms 400 > $O/main_queue.txt 2>&1
python tools/rocpd_gaps.py $DB --last-ms 400 --top 12 > $O/gaps.txt 2>&1
rm -rf $O/prof
cat $O/main_queue.txt; head -3 $O/gaps.txt; cut -c1-200 $O/bench_prof.json
