#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2l; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline.py -x -q -m gpu -k "spconv or conv" 2>&1 | tail -4
P="timeout 120 python tools/conv_probe.py --iters 10"
for shape in "0 96 96 k3" "0 128 96 k3" "1 96 96 k3" "0 32 32 k3" "1 32 32 k3" "1 32 64 k3" "1 128 96 k3" "0 96 96 up" "1 128 96 up"; do
  set -- $shape
  $P --level $1 --cin $2 --cout $3 --kind $4 2>&1 | grep sigma | tee -a $OUT/lowdens.txt
done
for s in 0.5 0.2 0.05; do $P --sigma $s --level 0 --cin 96 --cout 96 2>&1 | grep sigma | tee -a $OUT/lowdens.txt; $P --sigma $s --level 1 --cin 96 --cout 96 2>&1 | grep sigma | tee -a $OUT/lowdens.txt; done
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --all-variants > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
j=json.load(open("gpurun_out/r2l/bench.json")); r=j["roofline"]
print("steps/s %.2f ms %.2f"%(j["value"], j["ms_per_step"]), {k:(round(v["ms"]/10,2), round(v["tflops"],1)) for k,v in r["variants"].items()})
PY
