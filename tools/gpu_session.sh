# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4y}
R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
LIDIFF_PARITY_LOG=$O/parity_errors.jsonl timeout 2400 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python tools/parity_report.py $O/parity_errors.jsonl > $O/parity_errors.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-180 $O/bench_default.json
python bench.py --pipeline --scans 2 > $O/bench_pipeline.json 2>> $O/bench_default.err; cut -c1-300 $O/bench_pipeline.json
