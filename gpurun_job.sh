cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_golden.py -m gpu -x -q -k "tiger_assembly" 2>&1 | tail -12
