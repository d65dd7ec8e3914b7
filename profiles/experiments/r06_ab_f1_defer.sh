# k_flat1's deferred placement (VGX_F1_DEFER=1, default) against placing every segment as it is walked (=0), same box, round robin
for r in 1 2 3; do for v in 0 1; do
for c in cubics1m cubics1m_stroked; do
VGX_F1_DEFER=$v timeout 300 python bench.py --no-cpu --no-configs --config $c --steps 30 --warmup 5 --details /tmp/d.json 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=json.load(open('/tmp/d.json')); st=f['stage_ms']; print('defer=$v', '$c', d['ms_per_step'], {k:round(st[k],3) for k in st if k.startswith('flatten')})"
done; done; done
