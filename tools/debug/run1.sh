cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-closed-loop --no-coords-roofline 2>gpurun_out/b.err > gpurun_out/b_exec.json; tail -3 gpurun_out/b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_exec.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['fp32_equivalent_tflops'], r['executed'])
PY
