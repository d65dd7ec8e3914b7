"""Per-phase clock breakdown of k_flatten_build on the Tiger without instancing (needs a build with -DVGX_BUILD_PROFILE:
profiles/ab_variants.sh "bprof -DVGX_BUILD_PROFILE", then VGX_LIB=vg-renderer_amd/dbg/libvgx_bprof.so VGX_INST=0 VGX_TMPL=0 python profiles/build_phases.py)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VGX_INST", "0")
os.environ.setdefault("VGX_TMPL", "0")
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
st = dict(ctx.stage_times())
print({k: round(v, 3) for k, v in st.items()})
p = ctx.failure_info()["prof"]
waves, chunks = max(p[6], 1), max(p[4], 1)
us = [v / 100.0 for v in p[:6]]  # s_memtime: 100 MHz
print("waves %d chunks %d (%.1f per wave)" % (waves, chunks, chunks / waves))
print("per wave: alive %.1f us = records %.1f + walk %.1f + scans %.1f + placement %.1f + rest %.1f" % (
    us[5] / waves, us[0] / waves, us[1] / waves, us[2] / waves, us[3] / waves, (us[5] - sum(us[:4])) / waves))
print("per chunk: records %.2f us, walk %.2f us, scans + bookkeeping %.2f us, block switch + placement + records out %.2f us" % tuple(u / chunks for u in us[:4]))
