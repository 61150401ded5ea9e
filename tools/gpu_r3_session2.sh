#!/bin/bash
# Round-3 GPU session 2: tightened network tolerances, offset-grouped 'up' layers (probe A/B + bench A/B), timelines of the
# plain / skewed dense kernel (probe build), per-layer table of the bench.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s2
mkdir -p $O
export LIDIFF_PARITY_LOG=$PWD/$O/parity_errors.jsonl
rm -f $LIDIFF_PARITY_LOG
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_baseline.py -m gpu -q -k "not every_network_conv and not late_trajectory and not maps_bit_exact" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
U="3,256,256,up,-1,0;2,256,128,up,-1,0;1,128,96,up,-1,0;0,96,96,up,-1,0"
timeout 200 python tools/conv_probe.py --replicas 2 --cases "$U" > $O/probe_up_plain.txt 2>&1
timeout 200 python tools/conv_probe.py --replicas 2 --up-ordered --cases "$U" > $O/probe_up_ordered.txt 2>&1
timeout 200 python tools/conv_probe.py --timeline --replicas 1 --flags 0 --level 3 --cin 256 --cout 256 > $O/timeline_plain.txt 2>&1
timeout 200 python tools/conv_probe.py --timeline --replicas 1 --flags 8 --level 3 --cin 256 --cout 256 > $O/timeline_skew.txt 2>&1
timeout 200 python tools/conv_probe.py --timeline --replicas 2 --flags 0 --level 3 --cin 256 --cout 256 > $O/timeline_plain_r2.txt 2>&1
timeout 200 python tools/conv_probe.py --timeline --replicas 1 --sparse-hint 0 --level 2 --cin 128 --cout 128 > $O/timeline_s4_plain.txt 2>&1
timeout 200 python tools/conv_probe.py --timeline --replicas 1 --sparse-hint 1 --level 2 --cin 128 --cout 128 > $O/timeline_s4_packed.txt 2>&1
timeout 200 python tools/conv_probe.py --timeline --replicas 1 --level 3 --cin 128 --cout 128 > $O/timeline_s8_128.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --layer-table $O/layers_up_ordered.txt > $O/bench_up_ordered.json 2> $O/bench_up_ordered.err
LIDIFF_UP_ORDERED=0 timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --layer-table $O/layers_plain.txt > $O/bench_plain.json 2> $O/bench_plain.err
tail -3 $O/pytest.txt; grep -h sigma $O/probe_up_plain.txt $O/probe_up_ordered.txt | cut -c1-200; cut -c1-300 $O/bench_up_ordered.json; cut -c1-300 $O/bench_plain.json
