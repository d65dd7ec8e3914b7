for r in 1 2; do
for w in lane pool; do
VGX_WALK=$w timeout 300 python bench.py --no-cpu --no-configs --config tiger10k_command_parallel --steps 20 --warmup 5 --details /tmp/d.json 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=json.load(open('/tmp/d.json')); print('$w', d['ms_per_step'], 'flatten_build', round(f['stage_ms']['flatten_build'],3))"
done
done
