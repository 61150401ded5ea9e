#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/s42
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/tools/train_probe.py --steps 3 --precision bf16 > $O/train_prof.txt 2> $O/train_prof.err
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py $DB --top 45 > $O/train_kernel_stats.md 2>&1
rm -rf $O/prof
tail -5 $O/train_prof.txt
cut -c1-150 $O/train_kernel_stats.md | head -50
