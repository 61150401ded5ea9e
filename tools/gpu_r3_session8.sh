#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s8
mkdir -p $O
for CD in 0 1; do
for SG in 1.0 0.5; do
LIDIFF_CENTRE_DENSE=$CD timeout 300 python tools/conv_probe.py --replicas 2 --sigma $SG --centre-tail --cases "1,96,96,k3,1,0;0,96,96,k3,1,0;0,128,96,k3,1,0;1,32,32,k3,1,0;0,32,32,k3,1,0;1,32,64,k3,1,0" 2>&1 | grep -E "sigma|Error" | sed "s/$/ centre+tail centre_dense=$CD/" | cut -c1-240 >> $O/centre.txt
done; done
cat $O/centre.txt
