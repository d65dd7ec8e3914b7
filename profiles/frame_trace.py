"""The frame leg of bench.py (next_rows.frame_tiger_x1) alone, for a kernel trace: tests/golden/frame_tiger_x1.npz decoded once,
then N x vgx_tessellate with draw-command assembly armed. Usage: rocprofv3 --kernel-trace --stats -- python profiles/frame_trace.py [N] [noasm]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
rt = importlib.import_module("vg-renderer_amd.runtime")
cm = importlib.import_module("vg-renderer_amd.cmdlist")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
asm = not (len(sys.argv) > 2 and sys.argv[2] == "noasm")
fx = np.load(os.path.join(ROOT, "tests", "golden", "frame_tiger_x1.npz"))
kw = dict(mtx=[float(x) for x in fx["mtx"]], global_alpha=float(fx["global_alpha"]), tess_tol=float(fx["tess_tol"]), fringe=float(fx["fringe"]),
          canvas=(float(fx["canvas"][0]), float(fx["canvas"][1])), white_uv=[int(x) for x in fx["white_uv"]], font_image=int(fx["font_image"]))
rc, ps, draws, n = cm.decode(rt, fx["bytes"].tobytes(), **kw)
ctx = rt.Context(0)
pset = rt.PathSet(ctx, ps)
dd = rt.upload_draws(draws)
nd = int(draws.shape[0])
sz = rt.tessellate_count(ctx, pset, dd, nd)
dev = dd.device
bufs = rt.MeshBuffers(dev, sz["num_vertices"], sz["num_indices"], sz["num_meshes"])
cmds = torch.zeros((sz["num_meshes"] + 2) * 48, dtype=torch.uint8, device=dev)
ncmd = torch.zeros(1, dtype=torch.int64, device=dev)
uv = torch.zeros((sz["num_vertices"], 2), dtype=torch.int16, device=dev)
if asm:
    ctx.set_assembly(cmds, 65536, ncmd, split_state=True, uv=uv, uv_value=(int(fx["white_uv"][0]), 0))
for _ in range(5):
    rt.tessellate_async(ctx, pset, dd, nd, bufs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    rt.tessellate_async(ctx, pset, dd, nd, bufs)
torch.cuda.synchronize()
print("%.1f us per frame (%s)" % ((time.perf_counter() - t0) / N * 1e6, "assembly armed" if asm else "no assembly"))
