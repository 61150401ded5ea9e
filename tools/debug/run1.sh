cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_readfree.py -q -m gpu -x -k "two_piece or voided" 2>&1 | tail -5
LIDIFF_SPLIT_PIECES=2 python -m pytest tests/test_gpu_readfree.py tests/test_gpu_kernels.py -q -m gpu -x -k "two_piece or voided or split" 2>&1 | tail -3
