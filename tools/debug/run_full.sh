cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r06_pytest_gpu_summary.txt
python tools/parity_report.py > gpurun_out/r06_parity_errors.txt 2>&1
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
cut -c1-400 gpurun_out/r06_bench_default.json
