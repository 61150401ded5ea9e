#!/bin/bash
# rocprofv3 kernel trace of the training step (configs[4] per-GPU shape) -> per-kernel statistics + the kernel-class table.
#   usage: bash tools/gpu_train_profile.sh <tag> [bf16|32]
T=${1:-train}; P=${2:-bf16}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/tools/train_probe.py --steps 3 --warmup 1 --precision $P > $O/train_probe_$P.txt 2>&1
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py $DB --top 40 > $O/train_kernel_stats_$P.md 2>&1
python tools/train_classes.py $DB --steps 3 --warmup 1 --json $O/train_kernel_classes_$P.json > $O/train_kernel_classes_$P.txt 2>&1
rm -rf $O/prof
tail -2 $O/train_probe_$P.txt; cat $O/train_kernel_classes_$P.txt
