#!/bin/bash
# The multi-GPU numbers in ONE command, for the first box that has more than one MI355X (gpurun boxes have one: none of this
# has been measured -- DESIGN.md section 7 says "unmeasured" until a SCALE record exists).
#   bash tools/scale.sh [out_dir]          -> <out_dir>/scale_steps.jsonl   bench.py --gpus N, N = 1 2 4 8 (as many as visible):
#                                              one scan per GPU, weak scaling, no data-path collective; every line carries
#                                              rccl_ranks_seen (an all-reduce proves the ranks) and visible_devices
#                                             <out_dir>/scale_scans.jsonl   bench.py --gpus N --pipeline: whole scans sharded over ranks
#                                             <out_dir>/scale_tests.log     the >= 2-GPU tests (RCCL train loop + SyncBatchNorm)
# Nothing is extrapolated: a count the box cannot run is skipped and says so.
O=${1:-gpurun_out/scale}; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $HAVE" | tee $O/scale_README.txt
: > $O/scale_steps.jsonl; : > $O/scale_scans.jsonl
PORT=29611
for N in 1 2 4 8; do
  if [ "$N" -gt "$HAVE" ]; then echo "N=$N skipped: only $HAVE GPU(s) visible" | tee -a $O/scale_README.txt; continue; fi
  EXTRA="--no-cpu-baseline --no-train --no-alt --no-closed-loop"
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps 20 --warmup 5 $EXTRA >> $O/scale_steps.jsonl 2>> $O/scale_steps.err
    python bench.py --gpus 1 --pipeline --scans 2 >> $O/scale_scans.jsonl 2>> $O/scale_scans.err
  else
    PORT=$((PORT + 1))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $N --steps 20 --warmup 5 $EXTRA >> $O/scale_steps.jsonl 2>> $O/scale_steps.err
    PORT=$((PORT + 1))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $N --pipeline --scans 2 >> $O/scale_scans.jsonl 2>> $O/scale_scans.err
  fi
done
if [ "$HAVE" -ge 2 ]; then
  python -m pytest tests -m gpu -q -rs -k "two_rank or rccl or sync_batchnorm or bench_n_rank" > $O/scale_tests.log 2>&1
  tail -3 $O/scale_tests.log
fi
python - <<PY
import json
for name in ("scale_steps", "scale_scans"):
    rows = [json.loads(l) for l in open("$O/" + name + ".jsonl") if l.strip().startswith("{")]
    for r in rows:
        print(name, "n_gpus", r.get("n_gpus"), "ranks proven", r.get("rccl_ranks_seen"), r.get("metric"), r.get("value"), r.get("unit"))
PY
