mkdir -p gpurun_out/r4f
LIDIFF_PARITY_LOG=gpurun_out/r4f/parity.jsonl timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -m gpu -x -q -k "scatter or batch_norm or sync or training or two_rank or bit_reproducible" > gpurun_out/r4f/pytest.log 2>&1; tail -3 gpurun_out/r4f/pytest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-kernel-events --no-coords-roofline --no-closed-loop > gpurun_out/r4f/bench_train.json 2> gpurun_out/r4f/bench.err; wc -l gpurun_out/r4f/bench_train.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f/bench_train.json').readline())
for k in ('f32','bf16','bf16_syncbn'):
    print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in d['train'][k].items()})
PY
for c in "2,64,64,k3,-1,0" "2,128,128,k3,-1,0" "3,128,128,k3,-1,0" "3,256,256,k3,-1,0" "2,32,64,k3,-1,0"; do
  python tools/conv_probe.py --replicas 2 --timeline --cases "$c" >> gpurun_out/r4f/timeline.txt 2>&1
done
cat gpurun_out/r4f/timeline.txt | grep -v amdgpu.ids
bash tools/pmc_mfma.sh > gpurun_out/r4f/pmc.log 2>&1; tail -30 gpurun_out/r4f/pmc.log
