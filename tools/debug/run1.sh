cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "split" 2>&1 | tail -2
LIDIFF_SPLIT_PIECES=2 LIDIFF_PARITY_LOG=$PWD/gpurun_out/parity_f16x2.jsonl timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r06_pytest_gpu_f16x2_summary.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-closed-loop --no-coords-roofline 2>gpurun_out/b.err > gpurun_out/b_f16.json; tail -2 gpurun_out/b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_f16.json').read().strip().splitlines()[-1])
print('default', d['value'], d['ms_per_step'], 'native', d['native_fp32']['value'], 'f16x2', d['f16x2']['value'], d['f16x2']['ms_per_step'], d['f16x2'].get('roofline'))
PY
