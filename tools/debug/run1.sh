cd $GRAFT_REPO_ROOT
python tools/split3_table.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_split3.txt
timeout 900 python -m pytest tests/test_gpu_baseline.py -x -q -k "network_conv or cfg_pair" 2>&1 | tail -3
python tools/parity_report.py gpurun_out/parity_errors.jsonl > gpurun_out/r06_parity_errors_layers.txt 2>&1
