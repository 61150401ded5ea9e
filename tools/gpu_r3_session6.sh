#!/bin/bash
# hint sweep on the stride-4 / narrow layers (packed stages vs plain vs centre+tail), CFG pair stacked
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s6
mkdir -p $O
for SG in 1.0 0.5 0.2; do
C="2,64,64,k3,0,0;2,64,64,k3,1,0;2,32,64,k3,0,0;2,32,64,k3,1,0;2,64,128,k3,0,0;2,64,128,k3,1,0;1,96,96,k3,0,0;1,96,96,k3,1,0;1,32,32,k3,0,0;1,32,32,k3,1,0;0,96,96,k3,0,0;0,96,96,k3,1,0;0,32,32,k3,0,0;0,32,32,k3,1,0"
timeout 300 python tools/conv_probe.py --replicas 2 --sigma $SG --cases "$C" 2>&1 | grep sigma | cut -c1-200 >> $O/hint_sweep.txt
timeout 300 python tools/conv_probe.py --replicas 2 --sigma $SG --centre-tail --cases "2,64,64,k3,0,0;2,32,64,k3,0,0;2,128,128,k3,0,0;1,96,96,k3,0,0;1,32,32,k3,0,0;0,96,96,k3,0,0;0,32,32,k3,0,0" 2>&1 | grep sigma | sed 's/$/ centre+tail/' | cut -c1-220 >> $O/hint_sweep.txt
done
cat $O/hint_sweep.txt
