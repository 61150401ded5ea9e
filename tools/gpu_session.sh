# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4x}
R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -m gpu -x -q -k "fps or complete_scan or bench_n_rank" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --pipeline --scans 2 > $O/pipe_few.json 2> $O/err.txt
LIDIFF_FPS_GRID_ALL=1 python bench.py --pipeline --scans 2 > $O/pipe_all.json 2>> $O/err.txt
python - <<PY
import json
for n in ("few","all"):
    d=json.loads(open("$O/pipe_%s.json"%n).readline()); print(n, round(d["s_per_scan"],4), d["rank0"]["phases_s"])
PY
