cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -4; done
