#!/bin/bash
# Round-3 GPU session 4: accumulator-init flush + rulebook-built up orders: full GPU suite, probes, timelines, bench + layer table
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s4
mkdir -p $O
export LIDIFF_PARITY_LOG=$PWD/$O/parity_errors.jsonl
rm -f $LIDIFF_PARITY_LOG
timeout 1700 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
D="3,256,256,k3,-1,0;3,384,256,k3,-1,0;3,128,128,k3,-1,0;3,128,256,k3,-1,0;3,64,128,k3,-1,0;4,256,256,k3,-1,0;4,128,256,k3,-1,0;3,384,256,k1,-1,0"
S="2,128,128,k3,0,0;2,192,128,k3,0,0;2,192,128,k1,0,0;2,64,64,k3,-1,0;2,32,64,k3,-1,0;1,96,96,k3,-1,0;0,96,96,k3,-1,0;1,32,32,k3,-1,0"
timeout 300 python tools/conv_probe.py --replicas 2 --sigma 1.0 --cases "$D;$S" > $O/probe_sigma1.txt 2>&1
timeout 300 python tools/conv_probe.py --replicas 2 --sigma 0.2 --cases "3,256,256,k3,-1,0;4,256,256,k3,-1,0;2,128,128,k3,0,0;3,128,128,k3,-1,0;1,96,96,k3,-1,0;0,96,96,k3,-1,0" > $O/probe_sigma02.txt 2>&1
timeout 200 python tools/conv_probe.py --replicas 2 --up-ordered --cases "3,256,256,up,-1,0;2,256,128,up,-1,0;1,128,96,up,-1,0;0,96,96,up,-1,0" > $O/probe_up_ordered.txt 2>&1
for C in "3 256 256" "3 128 128" "2 128 128"; do set -- $C
  timeout 200 python tools/conv_probe.py --timeline --replicas 1 --sparse-hint 0 --level $1 --cin $2 --cout $3 2>&1 | grep -v amdgpu | cut -c1-250 >> $O/timelines.txt
done
timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --layer-table $O/layers.txt > $O/bench.json 2> $O/bench.err
LIDIFF_UP_ORDERED=0 timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events > $O/bench_noup.json 2> $O/bench_noup.err
timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --no-kernel-events > $O/bench_noevents.json 2> $O/bench_noevents.err
tail -3 $O/pytest.txt; grep -h sigma $O/probe_sigma1.txt $O/probe_up_ordered.txt | cut -c1-200; cat $O/timelines.txt; cut -c1-250 $O/bench.json; cut -c1-250 $O/bench_noup.json; cut -c1-250 $O/bench_noevents.json
