#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2o; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "centre_tail or low_density or epilogue" 2>&1 | tail -4
P="timeout 120 python tools/conv_probe.py --iters 10 --centre-tail"
for shape in "0 96 96 k3" "1 96 96 k3"; do
  set -- $shape
  $P --level $1 --cin $2 --cout $3 --kind $4 2>&1 | grep sigma | tee -a $OUT/lowdens.txt
done

timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --all-variants > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
j=json.load(open("gpurun_out/r2o/bench.json")); r=j["roofline"]
print("steps/s %.2f ms %.2f"%(j["value"], j["ms_per_step"]), {k:(round(v["ms"]/10,2), round(v["tflops"],1)) for k,v in r["variants"].items()})
PY
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2o/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r2o/prof.err; cd $GRAFT_REPO_ROOT; ls gpurun_out/r2o/prof | head; DB=$(find gpurun_out/r2o/prof -name '*.db' | head -1); python tools/rocpd_stats.py $DB --top 45 > gpurun_out/r2o/kernel_stats.md 2>&1; head -45 gpurun_out/r2o/kernel_stats.md
