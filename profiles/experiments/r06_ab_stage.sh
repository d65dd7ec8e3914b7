for v in head sidx sci sall; do
  export VGX_LIB=vg-renderer_amd/dbg/libvgx_$v.so
  echo "== $v"
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "round or Round or stroke or fuzz" 2>&1 | tail -2
  timeout 300 python -m pytest tests/test_gpu_fullsize_every_unit.py -x -q -m gpu -k "round_join_polylines" 2>&1 | tail -1
  for c in round10k tiger10k_round_ordinary; do
    timeout 300 python bench.py --no-cpu --no-configs --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['name'], d['ms_per_step'])"
    python -c "
import json; d=json.load(open('bench_details.json')); print({k:v for k,v in d['stage_ms'].items() if k in ('stroke_emit','fill_emit','mesh_prepare')})"
  done
done
