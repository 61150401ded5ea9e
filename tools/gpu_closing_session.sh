#!/bin/bash
# Closing GPU session of a round: full GPU suite, smoke, default bench, rocprofv3 kernel stats of the bench command.
set -u
cd "$(dirname "$0")/.."
R=$(pwd); OUT=$R/gpurun_out/closing; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -q -m gpu --durations=8 -rs 2>&1 | tail -30 > $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json; j=json.load(open('$OUT/bench_default.json')); r=j['roofline']
print('value %.2f ms %.2f frac %.3f step_frac %.3f alt %.2f cpu %.4f hbm %.4f narrow %.3f'%(j['value'], j['ms_per_step'], r['frac'], r['step_frac'], j['alt']['value'], j['cpu_baseline']['value'], j['roofline_hbm']['frac'], j['roofline_narrow']['frac']))"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt --no-coords-roofline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
cd $R
DB=$(find $OUT/prof -name '*.db' | head -1); python tools/rocpd_stats.py $DB --top 45 > $OUT/kernel_stats.md 2>&1; head -8 $OUT/kernel_stats.md; rm -rf $OUT/prof
# the launcher path of the metric on this box's one GPU: RCCL initialised by torch.distributed.run (world size 1), and the
# self-launching form refusing more ranks than there are devices
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-alt --no-coords-roofline > $OUT/bench_torchrun_n1.json 2> $OUT/bench_torchrun_n1.err; python -c "
import json; j=json.load(open('$OUT/bench_torchrun_n1.json')); print('torchrun n=1: value %.2f n_gpus %d rccl_ranks_seen %d'%(j['value'], j['n_gpus'], j['rccl_ranks_seen']))"
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_gpus2.out 2> $OUT/bench_gpus2.err; echo "bench --gpus 2 on a 1-GPU box: rc=$?"; grep -h "GPU(s) visible\|only" $OUT/bench_gpus2.err | tail -2
