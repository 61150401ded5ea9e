cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "split3" -x -q 2>&1 | tail -3
timeout 600 python tools/conv_probe.py --replicas 2 --kernel split3 --cases "3,256,256,k3,0,1;4,256,256,k3,0,1;3,128,128,k3,0,1;2,128,128,k3,0,1;3,384,256,k3,0,1" 2>&1 | grep avg_us
for a in 2 3; do echo "ablate=$a: $(LIDIFF_S3_ABLATE=$a timeout 600 python tools/conv_probe.py --replicas 2 --kernel split3 --cases '3,256,256,k3,0,1' 2>&1 | grep avg_us | sed 's/.*avg_us/avg_us/')"; done
