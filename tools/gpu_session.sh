# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4u}
R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
LIDIFF_PARITY_LOG=$O/parity.jsonl timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -x -q -k "mean or nn_match or maps_bit or golden or completion or overlapped or c1_one" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-train --no-closed-loop --no-coords-roofline"
python bench.py $B > $O/bench.json 2> $O/bench.err
python bench.py $B --no-kernel-events > $O/bench_noev.json 2>> $O/bench.err
python - <<PY
import json
for n in ("bench","bench_noev"):
    d=json.loads(open("$O/%s.json"%n).readline()); r=d.get("roofline",{})
    print(n, round(d["ms_per_step"],3), r.get("frac"), (r.get("serial") or {}).get("frac"))
    for k in ("roofline_narrow_hbm",):
        for l in (d.get(k) or {}).get("layers",[])[:3]: print("   ", l)
PY
bash tools/gpu_window.sh > /dev/null 2>&1; cp gpurun_out/window/window.txt $O/window.txt; head -75 $O/window.txt | tail -30
