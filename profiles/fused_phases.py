"""Per-phase wave-clock breakdown of the fused kernel (needs the profiling build: make -C vg-renderer_amd/csrc prof, then
VGX_LIB=vg-renderer_amd/libvgx_prof.so python profiles/fused_phases.py [instances]). Prints wave-microseconds per segment
and phase, and the wall time of the step."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

rt = importlib.import_module("vg-renderer_amd.runtime")
wl = importlib.import_module("vg-renderer_amd.workloads")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ps, d = wl.tiger(K)
ctx = rt.Context(0)
pset = rt.PathSet(ctx, ps)
dd = rt.upload_draws(d)
sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
for _ in range(3):
    rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
torch.cuda.synchronize()
t0 = time.perf_counter()
rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3
fi = ctx.failure_info()
names = ["ticket", "flatten", "meshes", "lookback", "table+fills", "strokes"]
segs = max(fi["prof"][6], 1)
print("status", fi["status"], "segment_items", fi["segment_items"], "segments", segs, "step ms %.3f" % ms)
tot = 0.0
for n, v in zip(names, fi["prof"]):
    us = v / 100.0  # 100 MHz
    tot += us
    print("%-12s %10.1f wave-ms total   %7.2f us / segment" % (n, us / 1e3, us / segs))
print("%-12s %10.1f wave-ms total   %7.2f us / segment" % ("sum", tot / 1e3, tot / segs))
chunks = max(fi["prof"][12], 1)
for n, v in zip(["decode", "walk", "bookkeeping", "store+records"], fi["prof"][8:12]):
    print("  flatten.%-14s %7.2f us / chunk  (%d chunks)" % (n, v / 100.0 / chunks, chunks))
