cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 250 python bench.py --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1 > gpurun_out/bench_d.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_d.json"))
print(d["value"], d["ms_per_step"], d["stage_ms"])
PY
