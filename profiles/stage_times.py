"""Tuning helper: per-kernel stage times of vgx_tessellate on Tiger x10k for a (possibly experimental) libvgx build.
Results are NOT checked (experimental builds may produce wrong output): VGX_LIB=path python profiles/stage_times.py"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

rt = importlib.import_module("vg-renderer_amd.runtime")
wl = importlib.import_module("vg-renderer_amd.workloads")


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "tiger"
    if which == "polylines":      # BASELINE config 4: 10k x 1000-segment polylines, strokeAA Round/Round
        ps, draws = wl.random_walk_polylines(n=10000, nseg=1000, seed=5678)
    elif which == "cubics":       # BASELINE config 2: 1 M independent cubics (stroked AA, Butt/Miter, so the stroker runs too)
        ps, draws = wl.random_cubics(1000000, seed=1234, box=1000.0)
        wl.set_stroke(draws, slice(None), 0xFF2080FF, 3.0, 0, 0, aa=True)
    else:
        K = int(which) if which.isdigit() else 10000
        ps, ops = wl.tiger_paths()
        draws = wl.tiger_draws(ops, K)
    n = draws.shape[0]
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(draws, 0)
    sizes = rt.tessellate_count(ctx, pset, dd, n)
    bufs = rt.MeshBuffers(torch.device("cuda", 0), sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    if os.environ.get("VGX_ASSEMBLE"):  # also run the draw-command assembly step (65536-vertex buffers)
        cap = 2 * (sizes["num_vertices"] // 65536) + 2
        cmds = torch.zeros(cap * 40, dtype=torch.uint8, device="cuda:0")
        ctx.set_assembly(cmds, 0, None)
    for _ in range(3):
        rt.tessellate_async(ctx, pset, dd, n, bufs)
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    acc = {}
    R = 5
    for _ in range(R):
        rt.tessellate_async(ctx, pset, dd, n, bufs)
        torch.cuda.synchronize()
        for k, v in ctx.stage_times():
            acc[k] = acc.get(k, 0.0) + v / R
    print(os.environ.get("VGX_LIB", "default"), which, "verts %d" % sizes["num_vertices"], "total %.3f" % sum(acc.values()), {k: round(v, 3) for k, v in acc.items()})


if __name__ == "__main__":
    main()
