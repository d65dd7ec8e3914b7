# the command scan's per-draw window records (VGX_DRAW_WIN=1, default) against the lookup through the draw's path (=0), same box, round robin
for r in 1 2 3; do for v in 0 1; do
for c in cubics1m tiger10k_command_parallel cubics1m_stroked; do
VGX_DRAW_WIN=$v timeout 300 python bench.py --no-cpu --no-configs --config $c --steps 30 --warmup 5 --details /tmp/d.json 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=json.load(open('/tmp/d.json')); st=f['stage_ms']; print('win=$v', '$c', d['ms_per_step'], {k:round(st[k],3) for k in st if k.startswith('flatten') or k.startswith('scan_cmd')})"
done; done; done
