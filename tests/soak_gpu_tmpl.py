"""GPU soak of the template mode (not collected by pytest): N random closed-shape drawings (every path command, serial shapes,
fills AA / plain / SSE index order, hairline and regular closed Miter strokes, many-small-mesh drawings that take the per-lane
fallback) x 32..90 instances under random affine transforms and colours -- every other seed in 2..9 classes (flavours of the drawing
with their own scales / tolerances / fill kinds / stroke widths) --, through vgx_tessellate against the reference oracle:
default tiles, small / odd tile sizes, with draw-command assembly armed on a third of the seeds (assembled indices against the
oracle's assembly). `python tests/soak_gpu_tmpl.py 300`."""
import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pyoracle
from util import assert_mesh_equal
rt = importlib.import_module("vg-renderer_amd.runtime"); wl = importlib.import_module("vg-renderer_amd.workloads")
pm = importlib.import_module("vg-renderer_amd.pathset")
ctxs = [rt.Context(0)]
for tile in ("64", "448", "1984"):
    os.environ["VGX_TMPL_TILE"] = tile
    ctxs.append(rt.Context(0))
os.environ.pop("VGX_TMPL_TILE")


def small_mesh_paths(seed, npaths):
    rs = np.random.RandomState(seed)
    b = pm.PathSetBuilder()
    for p in range(npaths):
        b.begin_path()
        for s in range(int(rs.randint(1, 8))):
            x, y = rs.uniform(-200, 200, size=2)
            if rs.uniform() < 0.4:
                b.rect(x, y, float(rs.uniform(2, 30)), float(rs.uniform(2, 30)))
            elif rs.uniform() < 0.5:
                b.circle(x, y, float(rs.uniform(1, 60)))
            else:
                b.move_to(x, y); b.line_to(x + float(rs.uniform(5, 20)), y + float(rs.uniform(-3, 3))); b.line_to(x + float(rs.uniform(-3, 3)), y + float(rs.uniform(5, 20))); b.close()
        b.end_path()
    return b.arrays()


bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for seed in range(7000, 7000 + n):
    rs = np.random.RandomState(seed)
    npaths = int(rs.randint(6, 80))
    ps = small_mesh_paths(seed, npaths) if seed % 5 == 0 else wl.closed_fuzz_paths(seed, npaths=npaths)
    ninst = int(rs.randint(32, 91))
    while ninst * npaths <= 2048:
        ninst += 13
    if seed % 4 == 2:  # general strokes: open sub-paths, every cap, Bevel joins, non-AA / hairline strokes
        ps = wl.fuzz_paths(seed, npaths=npaths, with_shapes=True, degenerate=bool(seed % 8 == 2))
        d = wl.template_general_draws(ps, seed, ninst)
    elif seed % 2:  # several classes: 2..9 flavours of the drawing, instances mixed at random
        d, _ = wl.template_class_draws(ps, seed, ninst, int(rs.randint(2, 10)))
    else:
        d = wl.template_draws(ps, seed, ninst, same_colors=bool(seed % 7 == 0))
    ctx = ctxs[seed % len(ctxs)]
    armed = seed % 3 == 0
    ref = pyoracle.tessellate(ps, d)
    pset = rt.PathSet(ctx, ps); dd = rt.upload_draws(d)
    max_vb = int(rs.choice([65536, 8192, 3000]))
    if armed:
        cmds = torch.zeros(100000 * 48, dtype=torch.uint8, device=dd.device); ncmd = torch.zeros(1, dtype=torch.int64, device=dd.device)
        d["state_key"] = np.repeat(rs.randint(0, 3, size=(d.shape[0] + 4) // 5), 5)[:d.shape[0]].astype(np.uint32)
        dd = rt.upload_draws(d)
        ctx.set_assembly(cmds, max_vb, ncmd, split_state=True)
    try:
        sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
        mode = ctx.failure_info()["segment_items"]
        bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
        rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs); torch.cuda.synchronize()
    finally:
        if armed:
            ctx.set_assembly(None)
    class G: pass
    g = G(); g.sizes = sizes
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    g.pos = bufs.pos[:nv].cpu().numpy(); g.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
    g.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16); g.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    try:
        assert mode == 5, ("not a template batch", mode)
        assert int(bufs.dev_status.item()) == 0, int(bufs.dev_status.item())
        if armed:
            big = (ref.meshes["num_vertices"] > max_vb).any()
            st, rcmds, ridx = pyoracle.assemble(ref.meshes, ref.idx, max_vb, mesh_keys=d["state_key"][ref.meshes["draw"]])
            assert st == 0 and not big
            ref.idx = ridx
            assert int(ncmd.item()) == rcmds.shape[0]
        assert_mesh_equal(g, ref, "tmpl soak %d" % seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, npaths, ninst, armed, str(e)[:300])
    pset.close()
print("template soak done: %d seeds, mismatches: %d" % (n, bad))
