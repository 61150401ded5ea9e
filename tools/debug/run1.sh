cd $GRAFT_REPO_ROOT/lidiff_amd/csrc
(cd $GRAFT_REPO_ROOT; python -m lidiff_amd.csrc.build --force > /dev/null 2>&1)
CASES="3,256,256,k3,-1,1;4,256,256,k3,-1,1;3,128,128,k3,-1,1;2,128,128,k3,-1,1"
for V in "-DLIDIFF_S3_CBW=2 -DLIDIFF_S3_NRQ=4" "-DLIDIFF_S3_CBW=4 -DLIDIFF_S3_NRQ=4" "-DLIDIFF_S3_CBW=4 -DLIDIFF_S3_NRQ=8" "-DLIDIFF_S3_CBW=2 -DLIDIFF_S3_NRQ=8"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip $V -c spconv_split3.hip -o spconv_split3.o 2>&1 | tail -3
  hipcc --offload-arch=gfx950 -shared -fPIC -o liblidiff_amd.so *.o 2>&1 | tail -3
  echo "== $V"
  (cd $GRAFT_REPO_ROOT; python tools/conv_probe.py --sigma 1.0 --replicas 2 --iters 30 --kernel split3 --cases "$CASES" 2>&1 | grep -v amdgpu.ids | grep -o "level=[0-9] kind=k3 [0-9]*->[0-9]*\|avg_us=.*" | paste - -)
done
