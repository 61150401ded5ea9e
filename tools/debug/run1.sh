cd $GRAFT_REPO_ROOT
python tools/debug/overlap_det.py 2>&1 | grep -v amdgpu | tail -5
