# cubics1m through vgx_flatten with other instance / bucket / grid choices of k_flat1 (same box)
for spec in "default" "VGX_F1_CAP=1024 VGX_F1_SEG=32" "VGX_F1_CAP=1024 VGX_F1_SEG=40" "VGX_F1_CAP=1664 VGX_F1_SEG=48" "VGX_F1_CAP=2048 VGX_F1_SEG=64" "VGX_F1_WAVES=1536" "VGX_F1_WAVES=3072" "VGX_F1_WAVES=4096" "VGX_F1_CAP=1024 VGX_F1_SEG=32 VGX_F1_WAVES=3072"; do
  if [ "$spec" = "default" ]; then E=""; else E="$spec"; fi
  env $E timeout 200 python bench.py --no-cpu --no-configs --config cubics1m --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$spec', d['ms_per_step'], d.get('ms_per_step_sustained'))"
done
