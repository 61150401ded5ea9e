#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2z; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline.py -q -m gpu -k "hash or maps or golden or nn_match or spconv_k3" --tb=short 2>&1 | tail -4 | tee $OUT/tests.txt
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
j=json.load(open("gpurun_out/r2z/bench.json"))
print("steps/s %.2f ms %.2f"%(j["value"], j["ms_per_step"]), "alt %.2f"%j["alt"]["value"], "hbm", j["roofline_hbm"]["ms"], j["roofline_hbm"]["frac"])
print(json.dumps(j["roofline_narrow"])[:1200])
PY
