"""GPU soak of template mode WITH draw-command assembly armed (not collected by pytest): N random drawings, periodic and static batches,
every stroke style (Round joins in two thirds of the seeds), random vertex-buffer sizes and state keys: vertex buffers, index buffer and
draw commands of the template path against the ordinary pipeline's (a context created with VGX_TMPL=0), byte for byte.
`python tests/soak_gpu_tmpl_asm.py 300`."""
import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
rt = importlib.import_module("vg-renderer_amd.runtime"); wl = importlib.import_module("vg-renderer_amd.workloads")
ctx_t = rt.Context(0)
os.environ["VGX_TMPL"] = "0"
ctx_o = rt.Context(0)
os.environ.pop("VGX_TMPL")


def run(ctx, ps, d, max_vb, split, static):
    ctx.set_static_batches(static)
    pset = rt.PathSet(ctx, ps); dd = rt.upload_draws(d)
    cmds = torch.zeros(400000 * 48, dtype=torch.uint8, device=dd.device)
    ncmd = torch.zeros(1, dtype=torch.int64, device=dd.device)
    ctx.set_assembly(cmds, max_vb, ncmd, split_state=split)
    try:
        sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
        mode = ctx.failure_info()["segment_items"]
        nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
        bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
        bufs.idx.fill_(-1)
        rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
        torch.cuda.synchronize()
    finally:
        ctx.set_assembly(None)
    st = int(bufs.dev_status.item()); n = int(ncmd.item())
    out = (st, mode, n, bufs.pos[:nv].clone(), bufs.color[:nv].clone(), bufs.idx[:ni].clone(), cmds[:n * 48].clone(), bufs.meshes[:nm * 32].clone())
    pset.close()
    return out


bad = 0
modes = {}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
base = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
for seed in range(base, base + n):
    rs = np.random.RandomState(seed)
    npaths = int(rs.randint(24, 97))
    ps = wl.closed_fuzz_paths(seed, npaths=npaths) if seed % 7 == 0 else wl.fuzz_paths(seed, npaths=npaths, with_shapes=bool(seed % 2), degenerate=False)
    ninst = int(rs.randint(33, 141))
    while ninst * npaths <= 2048:
        ninst += 17
    d = wl.template_general_draws(ps, seed, ninst, round_joins=(seed % 3 != 0))
    d["state_key"] = (np.arange(d.shape[0]) // int(rs.randint(5, 200))).astype(d["state_key"].dtype)
    static = seed % 2 == 1
    if static:
        d = d[rs.uniform(size=d.shape[0]) < 0.8]
        d = d[rs.permutation(d.shape[0])]
    max_vb = int(rs.choice([65536, 20000, 4096]))
    split = bool(seed % 3 == 1)
    a = run(ctx_t, ps, d, max_vb, split, static)
    b = run(ctx_o, ps, d, max_vb, split, False)
    modes[a[1]] = modes.get(a[1], 0) + 1
    ok = a[0] == b[0] and b[1] != 5
    if ok and a[0] == 0:
        ok = a[2] == b[2] and all(torch.equal(x, y) for x, y in zip(a[3:], b[3:]))
    if not ok:
        bad += 1; print("MISMATCH seed", seed, npaths, ninst, "static" if static else "periodic", "status", a[0], b[0], "modes", a[1], b[1], "cmds", a[2], b[2])
print("seeds", n, "mismatches", bad, "template-side count modes", modes)
sys.exit(1 if bad else 0)
