for r in 1 2 3; do for v in f1base f1nowait; do
VGX_LIB=vg-renderer_amd/dbg/libvgx_$v.so timeout 300 python bench.py --no-cpu --no-configs --config cubics1m --steps 30 --warmup 5 --details /tmp/d.json 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=json.load(open('/tmp/d.json')); print('$v', d['ms_per_step'], f['stage_ms'])"
done; done
