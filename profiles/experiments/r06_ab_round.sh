# same-box round-robin A/B of k_stroke_long variants on BASELINE configs[3]: bash profiles/experiments/r06_ab_round.sh v1 v2 ...
for r in 1 2 3; do
for v in "$@"; do
  VGX_LIB=vg-renderer_amd/dbg/libvgx_$v.so timeout 300 python bench.py --no-cpu --no-configs --config round10k --steps 30 --warmup 5 --details /tmp/d.json 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=json.load(open('/tmp/d.json')); print('$v', d['ms_per_step'], d.get('ms_per_step_sustained'), 'stroke_emit', round(f['stage_ms']['stroke_emit'],3))"
done
done
