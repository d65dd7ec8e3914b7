#!/bin/bash
# Builds the host-side decoder (vg-renderer_amd/csrc/vgx_cmdlist.hip: no device code) with ASan + UBSan and runs tests/asan_decoder_fuzz.py on it.
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/asan
sed -e 's/#include <hip\/hip_runtime.h>/#define __host__\n#define __device__/' -e "s#\"../../include/vgx.h\"#\"$PWD/include/vgx.h\"#" vg-renderer_amd/csrc/vgx_cmdlist.hip > /tmp/asan/vgx_cmdlist.cpp
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -I"$PWD/vg-renderer_amd/csrc" -x c++ /tmp/asan/vgx_cmdlist.cpp -o /tmp/asan/libcl_asan.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python tests/asan_decoder_fuzz.py
