#!/bin/bash
# ASan + UBSan + LeakSanitizer run of the per-call reference API on its HOST backend (test infrastructure, no GPU needed):
# vg-renderer_amd/host/vgx_host_backend.hip (the product's lane code compiled for the host) + host/vgx_compat.cpp +
# tests/compat_test.cpp (every vg::pathXXX / strokerXXX call against the oracle, concave fills against the reference's own
# strokerConcaveFillEndAA when oracle/_ref/libvgref.so is there).   bash tests/asan_compat_host.sh
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/asan_host; mkdir -p $OUT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
H=/opt/rocm/bin/hipcc
$H --offload-arch=gfx950 --cuda-host-only -O1 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $SAN -Wno-option-ignored -c vg-renderer_amd/host/vgx_host_backend.hip -o $OUT/hb.o
$H -O1 -std=c++17 -fPIC $SAN -Wno-option-ignored -c vg-renderer_amd/host/vgx_compat.cpp -o $OUT/compat.o
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -ffp-contract=off $SAN -o $OUT/compat_test tests/compat_test.cpp $OUT/compat.o $OUT/hb.o \
  -Iinclude -Ioracle -Ioracle/bx_shim -Ivg-renderer_amd/csrc -Lvg-renderer_amd -lvgx -Loracle -lvgoracle -L/opt/rocm/lib -lamdhip64 -ldl \
  -Wl,-rpath,$PWD/vg-renderer_amd -Wl,-rpath,$PWD/oracle -Wl,-rpath,/opt/rocm/lib
REF=$PWD/oracle/_ref/libvgref.so
[ -f $REF ] && export VGX_TEST_LIBVGREF=$REF
VGX_COMPAT_BACKEND=host $OUT/compat_test
