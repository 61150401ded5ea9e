# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4s}
R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
LIDIFF_PARITY_LOG=$O/parity.jsonl timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline.py -m gpu -x -q -k "spconv or conv_on or network_conv" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for s in 1.0 0.5 0.2; do for f in 64 0 128; do
  python tools/conv_probe.py --sigma $s --replicas 2 --iters 20 --cases "3,256,256,k3,0,$f;3,128,128,k3,0,$f;4,256,256,k3,0,$f;3,384,256,k3,0,$f;2,128,128,k3,0,$f" 2>&1 | grep TFLOP >> $O/packed64.txt
done; done
awk '{print $1,$2,$4,$8,$11,$12,$13}' $O/packed64.txt
