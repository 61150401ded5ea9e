#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/train_prof
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/tools/train_probe.py --steps 3 --precision bf16 > $O/train_bf16.log 2>&1
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py $DB --top 70 > $O/kernel_stats_bf16.md 2>&1
python tools/rocpd_gaps.py $DB --min-us 20 --top 15 --last-ms 300 > $O/gaps_bf16.txt 2>&1
tail -2 $O/train_bf16.log; head -60 $O/kernel_stats_bf16.md | cut -c1-170; head -25 $O/gaps_bf16.txt
rm -rf $O/prof
