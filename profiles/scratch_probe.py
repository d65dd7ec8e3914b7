"""Where does the per-process spread of the emit kernels come from: the caller's OUTPUT buffers or the context's SCRATCH (polyline
heap, mesh records)? Four contexts (each sizes its own scratch) x three output sets in one process, every pair timed on
Tiger x10k (stage times: flatten_build, fill_emit, stroke_emit, total). `python profiles/scratch_probe.py`"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rt = importlib.import_module("vg-renderer_amd.runtime"); wl = importlib.import_module("vg-renderer_amd.workloads")
ps, ops = wl.tiger_paths()
draws = wl.tiger_draws(ops, 10000)
dd = rt.upload_draws(draws)
ctxs, psets = [], []
for i in range(4):
    c = rt.Context(0); p = rt.PathSet(c, ps)
    sizes = rt.tessellate_count(c, p, dd, draws.shape[0])
    ctxs.append(c); psets.append(p)
outs = [rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]) for _ in range(3)]
for ci, (c, p) in enumerate(zip(ctxs, psets)):
    for oi, b in enumerate(outs):
        for _ in range(2):
            rt.tessellate_async(c, p, dd, draws.shape[0], b)
        torch.cuda.synchronize()
        c.set_profiling(True)
        acc = {}
        for _ in range(4):
            rt.tessellate_async(c, p, dd, draws.shape[0], b); torch.cuda.synchronize()
            for k, v in c.stage_times(): acc[k] = acc.get(k, 0.0) + v / 4
        c.set_profiling(False)
        print("context %d outputs %d: total %.3f flatten %.3f fill %.3f stroke %.3f" % (ci, oi, sum(acc.values()), acc["flatten_build"], acc["fill_emit"], acc["stroke_emit"]))
