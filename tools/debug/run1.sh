cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FAST="--no-cpu-baseline --no-coords-roofline --no-train --no-alt --no-closed-loop --no-kernel-events"
rm -rf $R/gpurun_out/prof_r06
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06 -o x -- python $R/bench.py --steps 20 --warmup 5 $FAST > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/bench_under_rocprof.err
cd $R; python tools/rocpd_stats.py gpurun_out/prof_r06/x_results.db --top 45 > gpurun_out/r06_kernel_stats.md
python tools/rocpd_main_queue.py gpurun_out/prof_r06/x_results.db > gpurun_out/r06_main_queue.txt 2>&1
rm -rf gpurun_out/prof_r06
