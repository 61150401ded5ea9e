#!/bin/bash
# One ad-hoc GPU session (edit per call): `gpurun -- 'bash tools/gpu_session.sh <tag>'` -> gpurun_out/<tag>/.
# The closing session of a round is tools/gpu_final.sh; this file is the scratch pad for A/B measurements in between, e.g.
#   python tools/conv_probe.py --replicas 2 --timeline --cases "3,256,256,k3,-1,0"      per-stage cycle budget of one layer
#   LIDIFF_<KNOB>=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train  a knob's fallback against the default
T=${1:-scratch}
R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
LIDIFF_PARITY_LOG=$O/parity_errors.jsonl timeout 2400 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-180 $O/bench_default.json
