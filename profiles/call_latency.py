"""Tuning helper: wall time per vgx_tessellate call for small batches (launch-latency bound regime)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

rt = importlib.import_module("vg-renderer_amd.runtime")
wl = importlib.import_module("vg-renderer_amd.workloads")

ctx = rt.Context(0)
for K in (1, 10, 100):
    ps, d = wl.tiger(K)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    bufs = rt.MeshBuffers(torch.device("cuda", 0), sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    for _ in range(5):
        rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    R = 200
    t0 = time.perf_counter()
    for _ in range(R):
        rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    t1 = time.perf_counter()
    for _ in range(R):
        rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
        torch.cuda.synchronize()
    dts = (time.perf_counter() - t1) / R
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    g.replay()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(R):
        g.replay()
    torch.cuda.synchronize()
    dtg = (time.perf_counter() - t2) / R
    print("tiger x%d: %.1f us per call back-to-back, %.1f us per call with sync, %.1f us per HIP-graph replay, %d verts" % (K, dt * 1e6, dts * 1e6, dtg * 1e6, sizes["num_vertices"]))
    del g
