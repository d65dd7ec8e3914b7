cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
VGX_LIB=$GRAFT_REPO_ROOT/vg-renderer_amd/dbg/libvgx_prev.so timeout 120 python profiles/stage_times.py 2>&1 | tail -1
timeout 120 python profiles/stage_times.py 2>&1 | tail -1
