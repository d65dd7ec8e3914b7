"""Tuning probe: whose placement decides the speed mode -- the context's scratch (polyline heap ...) or the caller's output
buffers? One context with four output sets, then one output set with four contexts."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rt = importlib.import_module("vg-renderer_amd.runtime"); wl = importlib.import_module("vg-renderer_amd.workloads")
ps, ops = wl.tiger_paths(); d = wl.tiger_draws(ops, 10000)
dev = torch.device("cuda", 0)


def timeit(ctx, pset, dd, bufs):
    for _ in range(2): rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize(); ctx.set_profiling(True)
    acc = {}
    for _ in range(5):
        rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs); torch.cuda.synchronize()
        for k, v in ctx.stage_times(): acc[k] = acc.get(k, 0.0) + v / 5
    return "total %.3f fill %.3f stroke %.3f" % (sum(acc.values()), acc["fill_emit"], acc["stroke_emit"])


ctxs, outs = [], []
for i in range(4):
    ctx = rt.Context(0); pset = rt.PathSet(ctx, ps); dd = rt.upload_draws(d, 0)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    ctxs.append((ctx, pset, dd))
    outs.append(rt.MeshBuffers(dev, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]))
for i in range(4):
    print("context 0, outputs %d: %s" % (i, timeit(*ctxs[0], outs[i])), flush=True)
for i in range(4):
    print("context %d, outputs 0: %s" % (i, timeit(*ctxs[i], outs[0])), flush=True)
