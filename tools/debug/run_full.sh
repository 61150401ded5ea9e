cd $GRAFT_REPO_ROOT
python -m lidiff_amd.csrc.build > /dev/null 2>&1
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r06_pytest_gpu_summary.txt
python tools/parity_report.py gpurun_out/parity_errors.jsonl > gpurun_out/r06_parity_errors.txt 2>&1
