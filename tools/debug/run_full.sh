cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r06_pytest_gpu_split3.txt
