# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4v}
R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
LIDIFF_PARITY_LOG=$O/parity.jsonl timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -x -q -k "cell_lists or nn_match or overlapped or completion_loop or c1_one or closed_loop" > $O/pytest.log 2>&1; tail -15 $O/pytest.log | cut -c1-250
B="--steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-train --no-closed-loop --no-coords-roofline"
for i in 1 2; do
python bench.py $B > $O/bench_cells$i.json 2> $O/bench.err
LIDIFF_MATCH_CELLS=0 python bench.py $B > $O/bench_brute$i.json 2>> $O/bench.err
done
python - <<PY
import json
for n in ("cells1","brute1","cells2","brute2"):
    d=json.loads(open("$O/bench_%s.json"%n).readline()); r=d.get("roofline",{})
    print(n, round(d["ms_per_step"],3), round(r.get("frac",0),4), round((r.get("serial") or {}).get("frac",0),4))
PY
bash tools/gpu_window.sh > /dev/null 2>&1; cp gpurun_out/window/window.txt $O/window.txt; grep -n "match\|cells\|spconv_thin\|mean_" $O/window.txt | head -30
