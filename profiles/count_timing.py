"""Host-timed cost of the sizing calls on a path set of a million paths: vgx_tessellate_count and vgx_partition
on 1 M one-cubic paths, AA strokes. `python profiles/count_timing.py`"""
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
rt = importlib.import_module("vg-renderer_amd.runtime"); wl = importlib.import_module("vg-renderer_amd.workloads")
ctx = rt.Context(0)
ps, d = wl.random_cubics(1000000, seed=1234, box=1000.0)
d = d.copy(); d["fill_flags"] = 0; d["stroke_flags"] = rt.capi.stroke_flags(rt.capi.CAP_BUTT, rt.capi.JOIN_MITER, aa=True); d["stroke_width"] = 4.0; d["stroke_color"] = 0xFF0000FF
pset = rt.PathSet(ctx, ps); dd = rt.upload_draws(d)
ts = []
for i in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("vgx_tessellate_count, 1 M one-cubic paths (stroke AA): ms per call", [round(x, 2) for x in ts], sizes["num_vertices"])
ts = []
for i in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    b, w = rt.partition(ctx, pset, dd, d.shape[0], 8)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("vgx_partition(8), same batch: ms per call", [round(x, 2) for x in ts], b[:3], w[:2])
