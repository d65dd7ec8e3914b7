#!/bin/bash
# Tuning helper: build the committed HEAD into vg-renderer_amd/dbg/libvgx_head.so so that one gpurun call can time
# HEAD and the working tree on the same box (box-to-box variation is larger than most single optimisations).
set -e
cd /root/repo
git stash -q
make -C vg-renderer_amd/csrc 2>&1 | grep -E "error" || true
cp vg-renderer_amd/libvgx.so vg-renderer_amd/dbg/libvgx_head.so
git stash pop -q
make -C vg-renderer_amd/csrc 2>&1 | grep -E "error" || true
