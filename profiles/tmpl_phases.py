"""Per-phase wall-clock breakdown of k_tmpl_emit (needs a build with -DVGX_TMPL_PROFILE: profiles/ab_variants.sh
"tprof -DVGX_TMPL_PROFILE", then VGX_LIB=vg-renderer_amd/dbg/libvgx_tprof.so python profiles/tmpl_phases.py)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

rt = importlib.import_module("vg-renderer_amd.runtime")
wl = importlib.import_module("vg-renderer_amd.workloads")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
JOIN = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # the strokes' LineJoin: 0 Miter (k_tmpl_emit), 2 Bevel (k_tmpl_emit_general), 1 Round (k_tmpl_emit_round)
ps, ops = wl.tiger_paths()
d = wl.tiger_draws(ops, K, join=JOIN)
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
print(dict(ctx.stage_times()))
p = ctx.failure_info()["prof"]
n = max(p[5], 1)
us = [v / 100.0 / n for v in p[:5]]  # wall_clock64: 100 MHz
print("workgroups %d; mean time since workgroup start (us): records in LDS %.2f | vertices in LDS %.2f | directions in LDS %.2f | stores issued %.2f | stores done %.2f" % (n, us[0], us[1], us[2], us[3], us[4]))
