#!/bin/bash
# Tuning helper: builds experimental variants of libvgx into vg-renderer_amd/dbg/ (one per line of VARIANTS: name + extra
# hipcc flags), so that ONE gpurun call can time them all on the same box with profiles/stage_times.py.
#   profiles/ab_variants.sh "nostore -DVGX_EXP_NOSTORE" "noload -DVGX_EXP_NOLOAD"
#   gpurun -- 'for v in head nostore noload; do VGX_LIB=vg-renderer_amd/dbg/libvgx_$v.so python profiles/stage_times.py; done'
set -e
cd "$(dirname "$0")/.."
SRC=vg-renderer_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
mkdir -p vg-renderer_amd/dbg
for spec in "$@"; do
  name=${spec%% *}; extra=${spec#* }; [ "$extra" = "$spec" ] && extra=""
  obj=vg-renderer_amd/dbg/obj_$name; mkdir -p $obj
  for f in vgx_api vgx_pathset vgx_tile vgx_flatten vgx_flat1 vgx_inst vgx_tmpl vgx_stroke vgx_concave vgx_merge vgx_cmdlist vgx_assemble vgx_cache; do
    slp=""; case $f in vgx_flatten|vgx_flat1|vgx_inst) slp="-fno-slp-vectorize";; esac
    /opt/rocm/bin/hipcc $FLAGS $slp $extra -c $SRC/$f.hip -o $obj/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vg-renderer_amd/dbg/libvgx_$name.so $obj/*.o
  echo "built vg-renderer_amd/dbg/libvgx_$name.so ($extra)"
done
