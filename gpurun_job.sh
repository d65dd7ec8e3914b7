cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
timeout 200 python profiles/stage_times.py 2>&1 | tail -1
