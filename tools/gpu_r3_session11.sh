#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s11
mkdir -p $O
for SG in 1.0 0.5; do
timeout 300 python tools/conv_probe.py --replicas 2 --sigma $SG --cases "2,128,128,k3,0,0;2,192,128,k3,0,0;2,64,64,k3,1,0;2,64,128,k3,0,0" 2>&1 | grep -E "sigma|Error" | cut -c1-220 >> $O/s4.txt
timeout 300 python tools/conv_probe.py --replicas 2 --sigma $SG --centre-tail --cases "2,128,128,k3,0,0;2,192,128,k3,0,0;2,64,64,k3,1,0;2,64,128,k3,0,0" 2>&1 | grep -E "sigma|Error" | sed "s/$/ centre+tail/" | cut -c1-240 >> $O/s4.txt
done
cat $O/s4.txt
