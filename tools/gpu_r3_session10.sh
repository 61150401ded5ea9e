#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/s10
mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -q -k "not every_network_conv and not late_trajectory and not maps_bit_exact and not gloo" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events 2>/dev/null | cut -c1-230; done
timeout 300 python tools/debug/step_timeline.py 2>&1 | grep -v amdgpu | head -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt --no-coords-roofline --no-train > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_gaps.py $DB --min-us 10 --top 12 --last-ms 300 > $O/gaps.txt 2>&1
head -16 $O/gaps.txt
rm -rf $O/prof
