cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "split3" 2>&1 | tail -2
NOEV="--no-kernel-events --no-cpu-baseline --no-train --no-alt --no-closed-loop --no-coords-roofline"
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 $NOEV 2>/dev/null | cut -c50-100; done
python tools/split3_table.py --sigmas 0.3,0.05 2>&1 | grep -v amdgpu.ids | cut -c1-150
