cd $GRAFT_REPO_ROOT
python tools/debug/train_ops.py 2>&1 | grep -v amdgpu.ids | head -70 | cut -c1-220
