cd $GRAFT_REPO_ROOT
python tools/split3_table.py --f16x2 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_f16x2_table.txt; wc -l gpurun_out/r06_f16x2_table.txt
python tools/parity_report.py gpurun_out/parity_f16x2.jsonl > gpurun_out/r06_parity_errors_f16x2.txt 2>&1; tail -3 gpurun_out/r06_parity_errors_f16x2.txt
