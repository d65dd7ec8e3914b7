"""GPU soak of template mode (not collected by pytest): N random drawings through vgx_tessellate against the reference oracle --
periodic batches (33..140 instances) and static batches (vgx_set_static_batches: the draws shuffled, a random part dropped), every
stroke style (Round joins in two thirds of the seeds; a quarter of the periodic seeds in several class flavours with Round joins), random affine transforms per instance, random tile sizes (read at vgx_create:
a few contexts), the steady-state call with OTHER transforms than the counted ones (Round joins: other sizes).
`python tests/soak_gpu_tmpl.py 200`."""
import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pyoracle
from util import assert_mesh_equal
rt = importlib.import_module("vg-renderer_amd.runtime"); wl = importlib.import_module("vg-renderer_amd.workloads")
ctxs = []
for tile in (None, "64", "192", "960"):
    if tile:
        os.environ["VGX_TMPL_TILE"] = tile
    ctxs.append(rt.Context(0))
    os.environ.pop("VGX_TMPL_TILE", None)
bad = 0
modes = {}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
base = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
for seed in range(base, base + n):
    rs = np.random.RandomState(seed)
    npaths = int(rs.randint(24, 97))
    closed_only = seed % 7 == 0
    ps = wl.closed_fuzz_paths(seed, npaths=npaths) if closed_only else wl.fuzz_paths(seed, npaths=npaths, with_shapes=bool(seed % 2), degenerate=bool(seed % 5 == 0))
    ninst = int(rs.randint(33, 141))
    while ninst * npaths <= 2048:
        ninst += 17
    d = wl.template_general_draws(ps, seed, ninst, round_joins=(seed % 3 != 0))
    static = seed % 2 == 1
    if not static and seed % 8 in (2, 6):  # round 6: Round joins in a template of several classes (64+ instances: the sizes pass per instance; fewer: the ordinary pipeline)
        ninst = max(ninst, 64 if seed % 16 == 2 else 40)
        d, _ = wl.template_class_round_draws(ps, seed, ninst, int(rs.randint(2, 9)), closed_aa_only=closed_only)
    if static:
        d = d[rs.uniform(size=d.shape[0]) < 0.8]
        d = d[rs.permutation(d.shape[0])]
    d2 = d.copy()
    if seed % 4 >= 2:  # steady-state call under other transforms / colours than the counted ones
        d2["mtx"] = rs.uniform(-2.0, 2.0, size=d2["mtx"].shape).astype(np.float32)
        d2["stroke_color"] = rs.randint(0, 1 << 32, size=d2.shape[0], dtype=np.uint64).astype(np.uint32)
    ctx = ctxs[seed % len(ctxs)]
    ctx.set_static_batches(static)
    ref = pyoracle.tessellate(ps, d2)
    pset = rt.PathSet(ctx, ps); dd = rt.upload_draws(d)
    rt.tessellate_count(ctx, pset, dd, d.shape[0])
    mode = ctx.failure_info()["segment_items"]
    modes[mode] = modes.get(mode, 0) + 1
    dd2 = rt.upload_draws(d2)
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
        assert_mesh_equal(g, ref, "tmpl soak %d" % seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, npaths, ninst, "static" if static else "periodic", "mode", mode, str(e)[:200], flush=True)
    pset.close()
    if (seed - base) % 100 == 99:  # (a run cut short by `timeout` still says how far it got)
        print("... seeds", seed - base + 1, "mismatches", bad, flush=True)
print("seeds", n, "mismatches", bad, "count modes", modes)
sys.exit(1 if bad else 0)
