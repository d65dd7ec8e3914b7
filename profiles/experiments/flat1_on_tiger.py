"""How long does vgx_flatten's one-walk kernel (k_flat1) take on the draw lists vgx_tessellate's ordinary pipeline flattens with
k_flatten_build / k_flatten_inst + k_flatten_gather? (Tiger x10k as 2.4 M unrelated draws; round10k's polylines.)
  python profiles/experiments/flat1_on_tiger.py"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

rt = importlib.import_module("vg-renderer_amd.runtime")
wl = importlib.import_module("vg-renderer_amd.workloads")
C = rt.C
capi = rt.capi


def run(name, ps, d, xform=True):
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    n = d.shape[0]
    z = capi.Sizes()
    rt._check(rt.lib().vgx_flatten_count(ctx.handle, pset.handle, dd.data_ptr(), n, C.byref(z), rt._stream_ptr()), "count")
    bufs = rt.FlatBuffers(dd.device, int(z.num_poly_vertices), int(z.num_subpaths), n)
    for _ in range(4):
        rt.flatten_async(ctx, pset, dd, n, bufs, xform)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    R = 10
    for _ in range(R):
        rt.flatten_async(ctx, pset, dd, n, bufs, xform)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / R * 1e3
    print(name, "draws", n, "poly verts", int(z.num_poly_vertices), "subpaths", int(z.num_subpaths), "ms/step %.3f" % ms, dict(ctx.stage_times()))
    pset.close()
    ctx.close()


ps, d = wl.tiger(10000)
run("tiger10k", ps, d)
ps, d = wl.random_walk_polylines(10000, 1000, seed=5678)
run("round10k", ps, d)
