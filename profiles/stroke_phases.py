"""Per-phase clock breakdown of k_stroke / k_stroke_long on BASELINE configs[3] (10 000 polylines x 1 000 points, Round joins): needs a
build with -DVGX_STROKE_PROFILE (profiles/ab_variants.sh "sprof -DVGX_STROKE_PROFILE", then
VGX_LIB=vg-renderer_amd/dbg/libvgx_sprof.so python profiles/stroke_phases.py)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

rt = importlib.import_module("vg-renderer_amd.runtime")
wl = importlib.import_module("vg-renderer_amd.workloads")
ps, d = wl.random_walk_polylines(10000, 1000, seed=5678)
ctx = rt.Context(0)
pset = rt.PathSet(ctx, ps)
dd = rt.upload_draws(d)
sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
for _ in range(3):
    rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
torch.cuda.synchronize()
ctx.set_profiling(True)
rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
torch.cuda.synchronize()
print({k: round(v, 3) for k, v in dict(ctx.stage_times()).items()})
p = ctx.failure_info()["prof"]
chunks = max(p[4], 1)
print("chunks %d; clocks per chunk: wait for the vertices %.0f, geometry + scans %.0f, emit + copy-out %.0f, all %.0f" % (
    chunks, p[0] / chunks, p[1] / chunks, p[2] / chunks, p[3] / chunks))
