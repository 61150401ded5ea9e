#!/bin/bash
# Round-2 closing session A: full GPU suite, default bench (with the CPU baseline), smoke().  Output under gpurun_out/r2fa/.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2fa; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -q -m gpu --durations=15 -rs 2>&1 | tail -60 > $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
