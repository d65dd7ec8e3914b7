"""GPU soak of k_flatten_thin (not collected by pytest): random path sets of moveTo / lineTo / close paths (vgx_thin.h) through
vgx_tessellate_count + vgx_tessellate against the reference oracle for a time budget -- set sizes from a frame to a few thousand
paths, with and without degenerate lineTo commands, draw lists that use every path once, some paths twice, or a shuffled subset,
the steady-state call under other transforms than the counted ones. `python -u tests/soak_gpu_thin.py 30 [first seed]`."""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pyoracle
from util import assert_mesh_equal
rt = importlib.import_module("vg-renderer_amd.runtime"); wl = importlib.import_module("vg-renderer_amd.workloads")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
base = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
ctx = rt.Context(0)
t0 = time.time(); bad = 0; n = 0; modes = {}
seed = base
while time.time() - t0 < budget:
    rs = np.random.RandomState(seed)
    npaths = int(rs.choice([30, 200, 700, 2200, 2600]))
    ps = wl.thin_fuzz_paths(seed, npaths=npaths, degenerate=bool(seed % 2))
    d = wl.fuzz_draws(ps, seed)
    kind = seed % 3
    if kind == 1:
        d = np.concatenate([d, d[rs.uniform(size=d.shape[0]) < 0.3]])
    elif kind == 2:
        d = d[rs.uniform(size=d.shape[0]) < 0.8]
    d = d[rs.permutation(d.shape[0])]
    d2 = d.copy()
    if seed % 4 >= 2:
        d2["mtx"] = rs.uniform(-2.0, 2.0, size=d2["mtx"].shape).astype(np.float32)
    ref = pyoracle.tessellate(ps, d2)
    pset = rt.PathSet(ctx, ps); dd = rt.upload_draws(d); dd2 = rt.upload_draws(d2)
    rt.tessellate_count(ctx, pset, dd, d.shape[0])
    mode = ctx.failure_info()["segment_items"]; modes[mode] = modes.get(mode, 0) + 1
    nv, ni, nm = int(ref.pos.shape[0]), int(ref.idx.shape[0]), int(ref.meshes.shape[0])
    bufs = rt.MeshBuffers(dd2.device, nv + 16, ni + 16, nm)
    rt.tessellate_async(ctx, pset, dd2, d2.shape[0], bufs); torch.cuda.synchronize()
    class G: pass
    g = G(); g.sizes = {"num_vertices": nv, "num_indices": ni, "num_meshes": nm}
    g.pos = bufs.pos[:nv].cpu().numpy(); g.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
    g.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16); g.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    try:
        assert int(bufs.dev_status.item()) == 0, "status %d" % int(bufs.dev_status.item())
        z = bufs.dev_sizes.cpu().numpy().view(np.uint64)
        assert (int(z[3]), int(z[4])) == (nv, ni), ("sizes", int(z[3]), int(z[4]), nv, ni)
        assert_mesh_equal(g, ref, "thin soak %d" % seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, npaths, kind, "mode", mode, str(e)[:200], flush=True)
    pset.close()
    n += 1; seed += 1
print("seeds", n, "from", base, "mismatches", bad, "flatten modes", modes, flush=True)
sys.exit(1 if bad else 0)
