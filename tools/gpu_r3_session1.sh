#!/bin/bash
# Round-3 GPU session 1: the whole GPU suite (parity errors recorded), A/B of the SKEW wave-half schedule and of packed stages
# on the stride-4 layers (conv_probe, CFG pair stacked as in the bench), the skew timeline, the default bench line.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s1
mkdir -p $O
export LIDIFF_PARITY_LOG=$PWD/$O/parity_errors.jsonl
rm -f $LIDIFF_PARITY_LOG
timeout 1700 python -m pytest tests -m gpu -q --durations=20 > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
D="3,256,256,k3,-1,0;3,256,256,k3,-1,8;3,384,256,k3,-1,0;3,384,256,k3,-1,8;3,128,128,k3,-1,0;3,128,128,k3,-1,8;3,128,256,k3,-1,0;3,128,256,k3,-1,8"
D="$D;4,256,256,k3,-1,0;4,256,256,k3,-1,8;4,128,256,k3,-1,0;4,128,256,k3,-1,8;3,256,256,up,-1,0;3,256,256,up,-1,8;3,384,256,k1,-1,0;3,384,256,k1,-1,8"
S="2,128,128,k3,0,0;2,128,128,k3,0,8;2,128,128,k3,1,0;2,192,128,k3,0,0;2,192,128,k3,0,8;2,192,128,k3,1,0;2,256,128,up,0,0;2,256,128,up,1,0;2,192,128,k1,0,0;2,192,128,k1,0,8"
timeout 300 python tools/conv_probe.py --replicas 2 --sigma 1.0 --cases "$D;$S" > $O/probe_sigma1.txt 2>&1
timeout 300 python tools/conv_probe.py --replicas 2 --sigma 0.2 --cases "3,256,256,k3,-1,0;3,256,256,k3,-1,8;4,256,256,k3,-1,0;4,256,256,k3,-1,8;2,128,128,k3,0,0;2,128,128,k3,0,8;2,128,128,k3,1,0" > $O/probe_sigma02.txt 2>&1
timeout 200 python tools/conv_probe.py --timeline --replicas 1 --flags 0 --level 3 --cin 256 --cout 256 > $O/timeline_plain.txt 2>&1
timeout 200 python tools/conv_probe.py --timeline --replicas 1 --flags 8 --level 3 --cin 256 --cout 256 > $O/timeline_skew.txt 2>&1
timeout 200 python tools/conv_probe.py --timeline --replicas 1 --sparse-hint 0 --level 2 --cin 128 --cout 128 > $O/timeline_s4_plain.txt 2>&1
timeout 200 python tools/conv_probe.py --timeline --replicas 1 --sparse-hint 1 --level 2 --cin 128 --cout 128 > $O/timeline_s4_packed.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
LIDIFF_CONV_FLAGS=8 timeout 300 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline > $O/bench_skew.json 2> $O/bench_skew.err
tail -3 $O/pytest.txt; cut -c1-220 $O/probe_sigma1.txt; cut -c1-400 $O/bench_skew.json
