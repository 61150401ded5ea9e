#!/bin/bash
# Round-2 GPU session A: dense-kernel parity + A/B probe, full GPU suite, bench A/B.  Output under gpurun_out/r2a/.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2a; mkdir -p $OUT
export PYTHONUNBUFFERED=1
echo "== quick dense-kernel parity" | tee $OUT/log.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spconv and not backward" 2>&1 | tail -15 | tee -a $OUT/log.txt
echo "== probe sweep: dense kernel" | tee -a $OUT/log.txt
timeout 300 python tools/conv_probe.py --sweep --iters 10 2>&1 | tee $OUT/probe_dense.txt | tail -25
echo "== probe sweep: tile kernels" | tee -a $OUT/log.txt
timeout 300 python tools/conv_probe.py --sweep --iters 10 --tile-kernel 2>&1 | tee $OUT/probe_tile.txt | tail -25
echo "== bench A/B" | tee -a $OUT/log.txt
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --all-variants > $OUT/bench_dense.json 2> $OUT/bench_dense.err; tail -c 600 $OUT/bench_dense.json
LIDIFF_CONV_TILE_KERNEL=1 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --all-variants > $OUT/bench_tile.json 2> $OUT/bench_tile.err; tail -c 600 $OUT/bench_tile.json
echo "== full GPU suite" | tee -a $OUT/log.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -45 | tee $OUT/pytest_gpu.txt
