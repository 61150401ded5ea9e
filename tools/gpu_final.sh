#!/bin/bash
# The closing GPU session of a round: the whole GPU suite with the achieved parity errors, the driver's bench command, the
# rocprofv3 kernel statistics / idle gaps / step-boundary window of that command, the PMC passes (traffic on the driver's command,
# MFMA occupancy per layer class), the per-layer table, the full-trajectory and whole-scan bench forms.
#   usage: bash tools/gpu_final.sh [tag]     -> gpurun_out/<tag>/ (copy what is to be judged into profiles/)
T=${1:-final}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
LIDIFF_PARITY_LOG=$O/parity_errors.jsonl timeout 2400 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python tools/parity_report.py $O/parity_errors.jsonl > $O/parity_errors.txt 2>&1
python bench.py --steps 20 --warmup 5 --layer-table $O/layer_table.txt > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
python bench.py --steps 20 --warmup 5 --no-kernel-events --no-cpu-baseline --no-train --no-alt --no-closed-loop --no-coords-roofline > $O/bench_no_events.json 2>> $O/bench_default.err
python bench.py --steps 50 --warmup 5 --cached-condition --no-cpu-baseline --no-train --no-closed-loop > $O/bench_steps50.json 2>> $O/bench_default.err
python bench.py --pipeline --scans 2 > $O/bench_pipeline.json 2>> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/bench.py --steps 20 --warmup 5 --no-kernel-events --no-cpu-baseline --no-train --no-alt --no-closed-loop --no-coords-roofline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py $DB --top 45 > $O/rocprofv3_kernel_stats.md 2>&1
python tools/rocpd_gaps.py $DB --last-ms 760 --top 15 > $O/idle_gaps.txt 2>&1
python tools/rocpd_main_queue.py $DB --last-ms 760 > $O/main_queue.txt 2>&1
python tools/rocpd_window.py $DB --nth 3 --ms 3.0 > $O/step_boundary_window.txt 2>&1
rm -rf $O/prof
PMC_STEPS=20 PMC_WARMUP=5 bash tools/pmc_bench.sh > $O/pmc_bench.log 2>&1; cp gpurun_out/pmc_bench/traffic.json $O/pmc_traffic.json
bash tools/pmc_mfma.sh > $O/pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma/summary.txt $O/pmc_mfma.txt; cp gpurun_out/pmc_mfma/summary.json $O/pmc_mfma.json
python tools/closed_loop_sensitivity.py 2>&1 | grep -v amdgpu > $O/closed_loop_sensitivity.txt
ls $O; head -5 $O/main_queue.txt; cat $O/closed_loop_sensitivity.txt
