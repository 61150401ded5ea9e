# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4k}
mkdir -p gpurun_out/$T
LIDIFF_PARITY_LOG=gpurun_out/$T/parity.jsonl timeout 1500 python -m pytest tests/test_gpu_network.py -m gpu -x -q -k "two_streams or overlapped or cfg_pair or completion_loop" > gpurun_out/$T/pytest.log 2>&1; tail -3 gpurun_out/$T/pytest.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-train --no-closed-loop --no-coords-roofline --no-kernel-events"
python bench.py $B > gpurun_out/$T/bench_stacked.json 2> gpurun_out/$T/bench.err
LIDIFF_CFG_STREAMS=1 python bench.py $B > gpurun_out/$T/bench_streams.json 2>> gpurun_out/$T/bench.err
LIDIFF_LAZY_XT=0 python bench.py $B > gpurun_out/$T/bench_stacked_nolazy.json 2>> gpurun_out/$T/bench.err
LIDIFF_CFG_STREAMS=1 python bench.py $B > gpurun_out/$T/bench_streams2.json 2>> gpurun_out/$T/bench.err
python - <<PY
import json
for n in ("stacked","streams","stacked_nolazy","streams2"):
    try:
        d=json.loads(open("gpurun_out/$T/bench_%s.json"%n).readline()); print(n, round(d["ms_per_step"],3))
    except Exception as e: print(n, "failed", e)
PY
tail -5 gpurun_out/$T/bench.err
