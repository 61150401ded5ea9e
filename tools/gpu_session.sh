# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4l}
mkdir -p gpurun_out/$T
LIDIFF_PARITY_LOG=gpurun_out/$T/parity.jsonl timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline.py -m gpu -x -q -k "spconv or conv or golden" > gpurun_out/$T/pytest.log 2>&1; tail -3 gpurun_out/$T/pytest.log
for c in "2,64,64,k3,-1,0" "2,128,128,k3,0,0" "3,256,256,k3,-1,0"; do
  python tools/conv_probe.py --replicas 2 --timeline --cases "$c" >> gpurun_out/$T/timeline.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/$T/timeline.txt | grep "TFLOP\|wave "
B="--steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-train --no-closed-loop --no-coords-roofline"
python bench.py $B > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
python bench.py $B --no-kernel-events > gpurun_out/$T/bench_noev.json 2>> gpurun_out/$T/bench.err
python - <<PY
import json
for n in ("bench","bench_noev"):
    d=json.loads(open("gpurun_out/$T/%s.json"%n).readline())
    r=d.get("roofline",{})
    print(n, round(d["ms_per_step"],3), r.get("frac"), (r.get("serial") or {}).get("frac"))
PY
