# same-box round-robin A/B of experimental builds on the headline batch (template mode) and on the tile kernel: bash profiles/experiments/r06_ab_headline.sh v1 v2 ...
for r in 1 2 3; do
for v in "$@"; do
  export VGX_LIB=vg-renderer_amd/dbg/libvgx_$v.so
  for c in tiger10k tiger10k_per_instance_flatten; do
    timeout 300 python bench.py --no-cpu --no-configs --config $c --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['name'], d['ms_per_step'], d.get('ms_per_step_sustained'))"
  done
done
done
