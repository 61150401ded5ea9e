cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_network.py -x -q -k "overlapped_coordinate or two_forward_mode" 2>&1 | tail -1; done
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r06_pytest_gpu_summary.txt
python tools/parity_report.py gpurun_out/parity_errors.jsonl > gpurun_out/r06_parity_errors.txt 2>&1
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
cut -c1-200 gpurun_out/r06_bench_default.json
python tools/split3_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_split3.txt; tail -3 gpurun_out/r06_split3.txt | cut -c1-160
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
LIDIFF_SPLIT_PIECES=2 LIDIFF_PARITY_LOG=$PWD/gpurun_out/parity_f16x2.jsonl timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/r06_pytest_gpu_f16x2_summary.txt
