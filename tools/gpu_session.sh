# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4o}
mkdir -p gpurun_out/$T
for s in 1.0 0.5 0.2; do
for f in 0 32; do
  python tools/conv_probe.py --sigma $s --replicas 2 --iters 20 --cases "2,128,128,k3,0,$f;2,192,128,k3,0,$f;3,128,128,k3,0,$f;3,64,128,k3,0,$f;3,256,256,k3,0,$f" 2>&1 | grep TFLOP >> gpurun_out/$T/tall.txt
done; done
cat gpurun_out/$T/tall.txt | cut -c1-200
