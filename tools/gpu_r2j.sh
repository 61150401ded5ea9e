#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2j; mkdir -p $OUT
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --all-variants > $OUT/bench_dense.json 2> $OUT/bench_dense.err; python - <<'PY'
import json
for n in ("dense","tile"):
    try:
        j=json.load(open(f"gpurun_out/r2j/bench_{n}.json")); r=j["roofline"]
        print(n, "steps/s %.2f ms %.2f bn128 %.1f TF frac %.3f conv_ms %.2f"%(j["value"], j["ms_per_step"], r["variants"]["bn128"]["tflops"], r["frac"], r["conv_ms_per_step_timed_variants"]))
    except Exception as e: print(n, "n/a", e)
PY
LIDIFF_CONV_TILE_KERNEL=1 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --all-variants > $OUT/bench_tile.json 2> $OUT/bench_tile.err
python - <<'PY'
import json
for n in ("dense","tile"):
    j=json.load(open(f"gpurun_out/r2j/bench_{n}.json")); r=j["roofline"]
    print(n, "steps/s %.2f ms %.2f bn128 %.1f TF frac %.3f conv_ms %.2f"%(j["value"], j["ms_per_step"], r["variants"]["bn128"]["tflops"], r["frac"], r["conv_ms_per_step_timed_variants"]))
PY
