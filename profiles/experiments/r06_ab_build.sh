# same-box round-robin A/B of k_flatten_build variants (profiles/ab_variants.sh builds them): bash profiles/experiments/r06_ab_build.sh v1 v2 ...
for r in 1 2 3; do
for v in "$@"; do
  export VGX_LIB=vg-renderer_amd/dbg/libvgx_$v.so
  for c in tiger10k_command_parallel; do
    timeout 300 python bench.py --no-cpu --no-configs --config $c --steps 20 --warmup 5 --details /tmp/d_$v.json 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=json.load(open('/tmp/d_$v.json')); print('$v', d['config']['name'], d['ms_per_step'], 'flatten_build', round(f['stage_ms']['flatten_build'],3))"
  done
done
done
