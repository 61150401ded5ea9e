cd $GRAFT_REPO_ROOT
FAST="--no-cpu-baseline --no-coords-roofline --no-train --no-alt --no-closed-loop"
for s in 1 0; do echo "== SPLIT3=$s"; LIDIFF_SPLIT3=$s timeout 900 python bench.py --steps 20 --warmup 5 $FAST --all-variants --layer-table gpurun_out/layer_table_split3_$s.txt 2>gpurun_out/bench_err_$s.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline'))"; done
LIDIFF_SPLIT3=1 timeout 1500 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -x -q -k "golden or batch2 or realistic or completion_loop or cfg_pair or c1 or network_conv or closed" 2>&1 | tail -8
