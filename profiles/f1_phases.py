"""Per-phase shader-clock ticks of k_flat1 (vgx_flatten, csrc/vgx_flat1.hip) on a flatten-only batch, from a -DVGX_F1_PROFILE build
(VGX_LIB=vg-renderer_amd/dbg/libvgx_f1prof.so). Usage: python profiles/f1_phases.py [cubics N box | tiger K]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rt = importlib.import_module("vg-renderer_amd.runtime")
wl = importlib.import_module("vg-renderer_amd.workloads")
kind = sys.argv[1] if len(sys.argv) > 1 else "cubics"
if kind == "cubics":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    box = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0
    ps, d = wl.random_cubics(n, seed=1234, box=box)
else:
    ps, d = wl.tiger(int(sys.argv[2]) if len(sys.argv) > 2 else 2000)
    d["fill_flags"] = 0; d["stroke_flags"] = 0
ctx = rt.Context(0)
pset = rt.PathSet(ctx, ps)
dd = rt.upload_draws(d)
r = rt.flatten(ctx, pset, dd, d.shape[0], entry="two_phase", to_host=False)
npv, nsp = r.sizes["num_poly_vertices"], r.sizes["num_subpaths"]
fb = rt.FlatBuffers(dd.device, npv, nsp, d.shape[0])
for _ in range(3):
    rt.flatten_async(ctx, pset, dd, d.shape[0], fb, apply_transform=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    rt.flatten_async(ctx, pset, dd, d.shape[0], fb, apply_transform=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
assert int(fb.dev_status.item()) == 0
fi = ctx.failure_info()
p = fi["prof"]
names = ["prologue(ticket,table,window,cmdrec)", "walk(root,tasks,loop)", "bookkeeping", "publish+lookback", "place+records", "walk steps", "chunks", "look-back spins", "look-back first poll ticks"]
tot = sum(p[:5]) or 1
print("%s: %.3f ms per call, %d poly verts, %.1f leaves/chunk" % (kind, ms, npv, npv / max(1, p[6])))
for i, nm in enumerate(names):
    if i < 5:
        print("  %-40s %14d ticks  %5.1f %%   %8.0f per chunk" % (nm, p[i], 100.0 * p[i] / tot, p[i] / max(1, p[6])))
    else:
        print("  %-40s %14d   (%.1f per chunk)" % (nm, p[i], p[i] / max(1, p[6])))
