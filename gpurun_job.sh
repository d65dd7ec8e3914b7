cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
VGX_LIB=$GRAFT_REPO_ROOT/vg-renderer_amd/dbg/libvgx_head.so timeout 120 python profiles/stage_times.py 2>&1 | tail -1 | sed 's/.*libvgx_head.so/HEAD/'
timeout 120 python profiles/stage_times.py 2>&1 | tail -1
done
