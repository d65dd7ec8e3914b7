one() { # name lib [env]
  VGX_LIB=vg-renderer_amd/dbg/libvgx_$2.so timeout 300 env $3 python bench.py --no-cpu --no-configs --config tiger10k_command_parallel --steps 20 --warmup 5 --details /tmp/d.json 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=json.load(open('/tmp/d.json')); print('$1', d['ms_per_step'], 'flatten_build', round(f['stage_ms']['flatten_build'],3))"
}
for r in 1 2; do
for v in "$@"; do one $v $v; done
done
