#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2u; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16" --tb=short 2>&1 | grep -v "^    \|^$" | tail -20 > $OUT/tests_kernels.txt; tail -1 $OUT/tests_kernels.txt
LIDIFF_SPLIT_PLANES=2 timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -q -m gpu -k "golden or batch2 or completion_loop or cfg_pair or fused_equals or c1_one or end_to_end or every_network_conv" --tb=short 2>&1 | grep -v "^    \|^$" | tail -30 > $OUT/tests_split2.txt; tail -3 $OUT/tests_split2.txt
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-coords-roofline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
j=json.load(open("gpurun_out/r2u/bench.json"))
print("steps/s %.2f ms %.2f"%(j["value"], j["ms_per_step"]), "alt", json.dumps(j.get("alt"))[:900])
PY
timeout 300 python tools/train_probe.py --steps 3 --precision 32 2>&1 | tail -1 | tee $OUT/train_32.txt
timeout 300 python tools/train_probe.py --steps 3 --precision bf16 2>&1 | tail -1 | tee $OUT/train_bf16.txt
