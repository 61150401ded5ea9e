cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "split" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-closed-loop --no-coords-roofline --layer-table gpurun_out/layer_k1up.txt 2>gpurun_out/b.err > gpurun_out/b_k1up.json; tail -2 gpurun_out/b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_k1up.json').read().strip().splitlines()[-1])
print('default', d['value'], d['ms_per_step'], 'native', d['native_fp32']['value'], 'f16x2', d['f16x2']['value'], d['f16x2']['ms_per_step'])
PY
grep -n "k=1 \|k=8 " gpurun_out/layer_k1up.txt | head -20
NOEV="--no-kernel-events --no-cpu-baseline --no-train --no-alt --no-closed-loop --no-coords-roofline"
for i in 1 2; do python bench.py --steps 20 --warmup 5 $NOEV 2>/dev/null | cut -c50-100; done
