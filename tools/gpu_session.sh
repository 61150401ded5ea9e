# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4p}
mkdir -p gpurun_out/$T
LIDIFF_PARITY_LOG=gpurun_out/$T/parity.jsonl timeout 1500 python -m pytest tests/test_gpu_baseline.py -m gpu -x -q -s -k "closed_loop" > gpurun_out/$T/pytest.log 2>&1; grep -v amdgpu gpurun_out/$T/pytest.log | tail -12
