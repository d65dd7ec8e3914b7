# same-box A/B of k_stroke_long: off (VGX_STROKE_LONG=0), three waves per SIMD, four (six spilled registers)
for r in 1 2; do
for v in off occ3 occ4; do
  if [ $v = off ]; then export VGX_LIB=vg-renderer_amd/dbg/libvgx_occ3.so VGX_STROKE_LONG=0; else export VGX_LIB=vg-renderer_amd/dbg/libvgx_$v.so VGX_STROKE_LONG=1; fi
  for c in round10k tiger10k_round_ordinary; do
    timeout 300 python bench.py --no-cpu --no-configs --config $c --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['name'], d['ms_per_step'])"
    python -c "
import json; d=json.load(open('bench_details.json')); print('   ', {k:v for k,v in d['stage_ms'].items() if k in ('stroke_emit','fill_emit')})"
  done
done
done
