cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
VGX_SERIAL_EMIT=1 timeout 120 python profiles/stage_times.py 2>&1 | tail -1 | sed 's/default/serial/'
timeout 120 python profiles/stage_times.py 2>&1 | tail -1
done
timeout 250 python bench.py --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1 | cut -c1-200
