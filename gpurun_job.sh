cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 120 python profiles/stage_times.py 2>&1 | tail -1
