# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4q}
R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
LIDIFF_PARITY_LOG=$O/parity.jsonl timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -m gpu -x -q -k "edge_cases or overlapped or pyramid or completion_loop" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-train --no-closed-loop --no-coords-roofline --no-kernel-events"
for i in 1 2; do
python bench.py $B > $O/bench_pinned$i.json 2> $O/bench.err
LIDIFF_PINNED_READ=0 python bench.py $B > $O/bench_pageable$i.json 2>> $O/bench.err
done
python - <<PY
import json
for n in ("pinned1","pageable1","pinned2","pageable2"):
    d=json.loads(open("$O/bench_%s.json"%n).readline()); print(n, round(d["ms_per_step"],3))
PY
PMC_STEPS=20 PMC_WARMUP=5 bash tools/pmc_bench.sh > $O/pmc_bench.log 2>&1; cp gpurun_out/pmc_bench/traffic.json $O/pmc_traffic.json; tail -3 $O/pmc_bench.log | cut -c1-150
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/bench.py --steps 20 --warmup 5 --no-kernel-events --no-cpu-baseline --no-train --no-alt --no-closed-loop --no-coords-roofline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py $DB --top 45 > $O/rocprofv3_kernel_stats.md 2>&1
python tools/rocpd_gaps.py $DB --last-ms 740 --top 15 > $O/idle_gaps.txt 2>&1
python tools/rocpd_main_queue.py $DB --last-ms 740 > $O/main_queue.txt 2>&1
python tools/rocpd_window.py $DB --nth 3 --ms 3.0 > $O/step_boundary_window.txt 2>&1
rm -rf $O/prof
head -9 $O/main_queue.txt; head -2 $O/idle_gaps.txt; cut -c1-160 $O/bench_under_rocprof.json
