#!/bin/bash
# Round-2 GPU session B: dense kernel after the nested-block restructure: parity, A/B probe, probe-build experiments.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2i; mkdir -p $OUT
export PYTHONUNBUFFERED=1
echo "== quick dense-kernel parity" | tee $OUT/log.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spconv and not backward" 2>&1 | tail -5 | tee -a $OUT/log.txt
P="timeout 120 python tools/conv_probe.py --iters 10"
for shape in "4 256 256"; do
  set -- $shape
  $P --level $1 --cin $2 --cout $3 2>&1 | grep sigma | sed 's/^/dense: /' | tee -a $OUT/ab.txt
  $P --level $1 --cin $2 --cout $3 --one-wave 2>&1 | grep sigma | sed 's/^/dense1:/' | tee -a $OUT/ab.txt
  $P --level $1 --cin $2 --cout $3 --tile-kernel 2>&1 | grep sigma | sed 's/^/tile:  /' | tee -a $OUT/ab.txt
done
echo "== probe build: issue priority (s_setprio) of the two waves of a SIMD, CB=1, level 3 256->256" | tee -a $OUT/log.txt
for m in 0 1 2 3; do
  pr=$((m * 65536))
  echo "-- prio mode $m (probe $pr): 0 none, 1 waves 0-3 high, 2 alternating per stage, 3 waves 4-7 high" | tee -a $OUT/probe.txt
  $P --level 3 --cin 256 --cout 256 --probe $pr --timeline 2>&1 | grep -v "amdgpu.ids\|shader clock\|timeline (" | tee -a $OUT/probe.txt
done
