#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/s13
mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_bf16_blocks.py tests/test_gpu_kernels.py -m gpu -q -k "training or train or bf16 or gloo or learn or backward" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
timeout 300 python tools/train_probe.py --steps 3 --precision bf16 2>&1 | tail -1
timeout 300 python tools/train_probe.py --steps 3 --precision 32 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/tools/train_probe.py --steps 3 --precision bf16 > $O/train_bf16.log 2>&1
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_gaps.py $DB --min-us 20 --top 12 --last-ms 300 > $O/gaps_bf16.txt 2>&1
head -30 $O/gaps_bf16.txt
rm -rf $O/prof
