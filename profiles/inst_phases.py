"""Per-phase wave-clock breakdown of k_flatten_inst (needs a build with -DVGX_INST_PROFILE:
profiles/ab_variants.sh "iprof -DVGX_INST_PROFILE", then VGX_LIB=vg-renderer_amd/dbg/libvgx_iprof.so python profiles/inst_phases.py)."""
import importlib
import os
import sys

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
ctx.set_profiling(True)
rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
torch.cuda.synchronize()
print(dict(ctx.stage_times()))
fi = ctx.failure_info()
p = fi["prof"]
waves = int(os.environ.get("VGX_INST_WAVES", "4096"))
us = [v / 100.0 for v in p[:4]]
tasks, cubics = max(p[4], 1), max(p[5], 1)
print("waves %d tasks %d cubics %d" % (waves, tasks, cubics))
print("per wave: alive %.1f us | prologues %.1f us | command loops %.1f us (cubic walks %.1f us)" % (us[3] / waves, us[0] / waves, us[2] / waves, us[1] / waves))
print("per task: prologue %.2f us, command loop %.2f us | per cubic walk %.3f us" % (us[0] / tasks, us[2] / tasks, us[1] / cubics))
