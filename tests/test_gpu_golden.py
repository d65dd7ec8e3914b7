"""GPU: the HIP path against the committed golden vectors (generated from the reference's own sources) and
full-size property checks that do not need the oracle to scale."""
import importlib

import numpy as np
import pytest

import golden_util as gu
from util import assert_flat_equal, assert_mesh_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    return importlib.import_module("vg-renderer_amd.runtime")


def _mesh(rt, ctx, ps, d):
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    r = rt.tessellate(ctx, pset, dd, d.shape[0])
    pset.close()
    return r


def test_known_answers_on_gpu(rt, gpu_ctx, vgr):
    ps = gu.zigzag_set()
    recs = [r for r in gu.known_answers() if r["mode"] in ("aa", "plain", "thin")]
    draws = np.concatenate([gu.known_answer_draw(vgr, r) for r in recs])
    got = _mesh(rt, gpu_ctx, ps, draws)
    assert got.sizes["num_meshes"] == len(recs)
    for m, rec in zip(got.meshes, recs):
        assert (int(m["num_vertices"]), int(m["num_indices"])) == (rec["verts"], rec["idx"]), rec
        v0, i0 = int(m["first_vertex"]), int(m["first_index"])
        assert gu.sha(got.idx[i0:i0 + rec["idx"]]) == rec["idx_sha"], rec
        assert gu.sha(got.color[v0:v0 + rec["verts"]]) == rec["col_sha"], rec
        assert gu.sha(got.pos[v0:v0 + rec["verts"]]) == rec["pos_sha"], rec


@pytest.mark.parametrize("seed", [7, 8])
def test_golden_fuzz_on_gpu(rt, gpu_ctx, seed):
    ps, draws, g = gu.load_fuzz(seed)
    got = _mesh(rt, gpu_ctx, ps, draws)
    assert_mesh_equal(got, g, "golden fuzz %d" % seed)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(draws)
    f = rt.flatten(gpu_ctx, pset, dd, draws.shape[0], apply_transform=True)
    assert_flat_equal(f, g, "golden fuzz %d polyline" % seed)
    f = rt.flatten(gpu_ctx, pset, dd, draws.shape[0], apply_transform=False)
    assert np.array_equal(f.poly.view(np.uint32), g.poly_raw.view(np.uint32))
    pset.close()


@pytest.mark.parametrize("name", ["config0_single_cubic", "tiger_x1", "tiger_x3", "polylines_round_round_20x300", "cubics_2000_box1000"])
def test_golden_checksums_on_gpu(rt, gpu_ctx, wl, name):
    ps, d = gu.workload_by_name(wl, name)
    got = _mesh(rt, gpu_ctx, ps, d)
    c = gu.checksums()[name]
    for k in ("num_poly_vertices", "num_subpaths", "num_meshes", "num_vertices", "num_indices"):
        assert got.sizes[k] == c["sizes"][k], (name, k)
    for k in ("pos", "color", "idx", "meshes"):
        assert gu.sha(getattr(got, k)) == c[k], (name, k)


def test_full_size_tiger_properties(rt, gpu_ctx, wl, oracle, monkeypatch):
    """BASELINE config 2 at full size (Tiger x10k = 2.4 M draws, 415 M vertices): size-independent checks.
      - totals = 10 000 x the single-instance totals of the golden run (flatten is translation invariant),
      - mesh table is a consistent exclusive scan,
      - every instance's colour stream and the index streams of its convex-fill meshes are IDENTICAL to
        instance 0's (mesh-local indices, closed-form in N); stroke indices may legitimately differ where the
        translation's rounding flips a join's inner side (dot >= 0 test, stroker.cpp:1534-1535),
      - randomly sampled instances match the oracle bit-exactly, positions included."""
    import torch
    K = 10000
    ps, ops = wl.tiger_paths()
    draws = wl.tiger_draws(ops, K)
    one = gu.checksums()["tiger_x1"]["sizes"]
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(draws)
    r = rt.tessellate(gpu_ctx, pset, dd, draws.shape[0], to_host=False)
    for k in ("num_poly_vertices", "num_subpaths", "num_meshes", "num_vertices", "num_indices"):
        assert r.sizes[k] == K * one[k], k
    nv1, ni1, nm1 = one["num_vertices"], one["num_indices"], one["num_meshes"]
    idx = r.bufs.idx[:K * ni1].view(K, ni1)
    col = r.bufs.color[:K * nv1].view(K, nv1)
    assert bool((col == col[0:1]).all().item())
    meshes = r.bufs.meshes[:K * nm1 * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    fill_mask = np.zeros(ni1, dtype=bool)
    for m in meshes[:nm1]:
        if (int(m["subpath_kind"]) >> 28) == rt.capi.MESH_FILL_AA:
            fill_mask[int(m["first_index"]):int(m["first_index"]) + int(m["num_indices"])] = True
    fm = torch.from_numpy(fill_mask).to(idx.device)
    assert bool((idx[:, fm] == idx[0:1, fm]).all().item())
    frac_diff = float((idx != idx[0:1]).float().mean().item())
    assert frac_diff < 1e-3, frac_diff
    assert np.array_equal(meshes["first_vertex"][1:], np.cumsum(meshes["num_vertices"].astype(np.uint64))[:-1])
    assert np.array_equal(meshes["first_index"][1:], np.cumsum(meshes["num_indices"].astype(np.uint64))[:-1])
    assert np.array_equal(meshes["draw"], np.repeat(np.arange(K * len(ops), dtype=np.uint32), np.tile(np.bincount(meshes["draw"][:nm1], minlength=len(ops)), K)))
    rs = np.random.RandomState(1)
    for inst in [0, K - 1] + [int(x) for x in rs.randint(1, K - 1, size=6)]:
        ref = oracle.tessellate(ps, draws[inst * len(ops):(inst + 1) * len(ops)])
        pos = r.bufs.pos[inst * nv1:(inst + 1) * nv1].cpu().numpy()
        assert np.array_equal(pos.view(np.uint32), ref.pos.view(np.uint32)), inst
        assert np.array_equal(idx[inst].cpu().numpy().view(np.uint16), ref.idx), inst
        assert np.array_equal(col[inst].cpu().numpy().view(np.uint32), ref.color), inst
    # The entry point bench.py times is the ASYNCHRONOUS single call (vgx_tessellate: single-pass k_flatten_build with
    # its ~4 heap-block switches per wave at this size, no host round trip), not the count + emit pair above. Its streams
    # must be byte-identical to the two-phase result, and the same sampled instances must match the reference oracle
    # on the async buffers themselves.
    b2 = rt.MeshBuffers(dd.device, r.sizes["num_vertices"], r.sizes["num_indices"], r.sizes["num_meshes"])
    for _ in range(2):  # second call = steady state (scratch already sized)
        rt.tessellate_async(gpu_ctx, pset, dd, draws.shape[0], b2)
    torch.cuda.synchronize()
    assert int(b2.dev_status.item()) == 0
    got = b2.dev_sizes.cpu().numpy()
    assert int(got[3]) == r.sizes["num_vertices"] and int(got[4]) == r.sizes["num_indices"] and int(got[2]) == r.sizes["num_meshes"]
    nv, ni, nm = r.sizes["num_vertices"], r.sizes["num_indices"], r.sizes["num_meshes"]
    assert torch.equal(b2.pos[:nv].view(torch.int32), r.bufs.pos[:nv].view(torch.int32))
    assert torch.equal(b2.color[:nv], r.bufs.color[:nv])
    assert torch.equal(b2.idx[:ni], r.bufs.idx[:ni])
    assert torch.equal(b2.meshes[:nm * 32], r.bufs.meshes[:nm * 32])
    rs = np.random.RandomState(2)
    for inst in [0, K - 1] + [int(x) for x in rs.randint(1, K - 1, size=6)]:
        ref = oracle.tessellate(ps, draws[inst * len(ops):(inst + 1) * len(ops)])
        assert np.array_equal(b2.pos[inst * nv1:(inst + 1) * nv1].cpu().numpy().view(np.uint32), ref.pos.view(np.uint32)), inst
        assert np.array_equal(b2.idx[inst * ni1:(inst + 1) * ni1].cpu().numpy().view(np.uint16), ref.idx), inst
        assert np.array_equal(b2.color[inst * nv1:(inst + 1) * nv1].cpu().numpy().view(np.uint32), ref.color), inst
    del idx, col, b2
    torch.cuda.empty_cache()
    # The same batch through the command-parallel single-pass kernel (k_flatten_build: what every batch that is NOT
    # instanced runs, with its heap-block switches at 140 M polyline vertices): byte-identical streams.
    monkeypatch.setenv("VGX_INST", "0")
    ctx0 = rt.Context(0)
    pset0 = rt.PathSet(ctx0, ps)
    rt.tessellate_count(ctx0, pset0, dd, draws.shape[0])
    b3 = rt.MeshBuffers(dd.device, nv, ni, nm)
    ctx0.set_profiling(True)
    rt.tessellate_async(ctx0, pset0, dd, draws.shape[0], b3)
    torch.cuda.synchronize()
    assert "flatten_build" in [n for n, _ in ctx0.stage_times()]
    assert int(b3.dev_status.item()) == 0
    assert torch.equal(b3.pos[:nv].view(torch.int32), r.bufs.pos[:nv].view(torch.int32))
    assert torch.equal(b3.color[:nv], r.bufs.color[:nv])
    assert torch.equal(b3.idx[:ni], r.bufs.idx[:ni])
    assert torch.equal(b3.meshes[:nm * 32], r.bufs.meshes[:nm * 32])
    pset0.close()
    ctx0.close()
    del r, b3
    torch.cuda.empty_cache()
    pset.close()


def test_full_size_tiger_varied_scales(rt, wl, oracle, monkeypatch):
    """bench.py's tiger10k_varied at full size (every instance at one of 7 scales -- 18 distinct avgScale values -- and its own
    rotation: 2.4 M draws, ~0.63 G vertices). Three pipelines, byte for byte the same streams: template mode with one template
    per class (mode 5, the default), the instanced kernel's periodic mapping with the instances sorted by tolerance class
    (VGX_TMPL_CLASSES=0: mode 4) and the command-parallel kernel (VGX_INST=0: k_flatten_build, an independent implementation of
    the flatten); instance by instance for a sample, the reference oracle."""
    import torch
    K = 10000
    ps, ops = wl.tiger_paths()
    draws = wl.tiger_varied_draws(ops, K)
    P = len(ops)
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(draws)
    sizes = rt.tessellate_count(ctx, pset, dd, draws.shape[0])
    assert ctx.failure_info()["segment_items"] == 5
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    b = rt.MeshBuffers(dd.device, nv, ni, nm)
    for _ in range(2):
        rt.tessellate_async(ctx, pset, dd, draws.shape[0], b)
    torch.cuda.synchronize()
    assert int(b.dev_status.item()) == 0
    got = b.dev_sizes.cpu().numpy()
    assert int(got[3]) == nv and int(got[4]) == ni and int(got[2]) == nm
    meshes = b.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    assert np.array_equal(meshes["first_vertex"][1:], np.cumsum(meshes["num_vertices"].astype(np.uint64))[:-1])
    assert np.array_equal(meshes["first_index"][1:], np.cumsum(meshes["num_indices"].astype(np.uint64))[:-1])
    first_mesh_of_draw = np.searchsorted(meshes["draw"], np.arange(K * P + 1, dtype=np.uint64), side="left")
    rs = np.random.RandomState(3)
    for inst in [0, K - 1] + [int(x) for x in rs.randint(1, K - 1, size=6)]:
        ref = oracle.tessellate(ps, draws[inst * P:(inst + 1) * P])
        m0, m1 = int(first_mesh_of_draw[inst * P]), int(first_mesh_of_draw[(inst + 1) * P])
        assert m1 - m0 == ref.meshes.shape[0], inst
        v0, i0 = int(meshes["first_vertex"][m0]), int(meshes["first_index"][m0])
        v1, i1 = v0 + ref.pos.shape[0], i0 + ref.idx.shape[0]
        assert np.array_equal(b.pos[v0:v1].cpu().numpy().view(np.uint32), ref.pos.view(np.uint32)), inst
        assert np.array_equal(b.idx[i0:i1].cpu().numpy().view(np.uint16), ref.idx), inst
        assert np.array_equal(b.color[v0:v1].cpu().numpy().view(np.uint32), ref.color), inst
    for var, mode in (("VGX_TMPL_CLASSES", 4), ("VGX_INST", 0)):
        monkeypatch.setenv(var, "0")
        ctx0 = rt.Context(0)
        monkeypatch.delenv(var)
        pset0 = rt.PathSet(ctx0, ps)
        s0 = rt.tessellate_count(ctx0, pset0, dd, draws.shape[0])
        assert ctx0.failure_info()["segment_items"] == mode, (var, ctx0.failure_info()["segment_items"])
        assert (s0["num_vertices"], s0["num_indices"], s0["num_meshes"]) == (nv, ni, nm)
        b0 = rt.MeshBuffers(dd.device, nv, ni, nm)
        rt.tessellate_async(ctx0, pset0, dd, draws.shape[0], b0)
        torch.cuda.synchronize()
        assert int(b0.dev_status.item()) == 0
        assert torch.equal(b0.pos[:nv].view(torch.int32), b.pos[:nv].view(torch.int32)), var
        assert torch.equal(b0.color[:nv], b.color[:nv]), var
        assert torch.equal(b0.idx[:ni], b.idx[:ni]), var
        assert torch.equal(b0.meshes[:nm * 32], b.meshes[:nm * 32]), var
        pset0.close(); ctx0.close()
        del b0
        torch.cuda.empty_cache()
    pset.close(); ctx.close()
    del b
    torch.cuda.empty_cache()


def test_full_size_flatten_1m_cubics(rt, gpu_ctx, wl, oracle):
    """BASELINE config 1: 1 M independent cubics, flatten only. Oracle on a 20 k sample + global properties."""
    n = 1000000
    ps, d = wl.random_cubics(n, seed=1234, box=1000.0)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    f = rt.flatten(gpu_ctx, pset, dd, n, apply_transform=False)
    di = f.draw_info
    assert int(di["num_poly_vertices"].sum()) == f.sizes["num_poly_vertices"]
    assert np.array_equal(di["first_poly_vertex"][1:], np.cumsum(di["num_poly_vertices"].astype(np.uint64))[:-1])
    assert int(di["num_subpaths"].min()) == 1 and int(di["num_subpaths"].max()) == 1
    # first vertex of every path is its moveTo point, last vertex its cubic end point
    first = f.poly[di["first_poly_vertex"].astype(np.int64)]
    last = f.poly[(di["first_poly_vertex"] + di["num_poly_vertices"] - 1).astype(np.int64)]
    pts = ps.args.reshape(n, 8)
    assert np.array_equal(first, pts[:, 0:2]) and np.array_equal(last, pts[:, 6:8])
    sel = slice(500000, 520000)
    sub_ps = vgr_subset(ps, 500000, 520000)
    ref = oracle.flatten(sub_ps, d[:20000], apply_transform=False)
    a = int(di["first_poly_vertex"][500000])
    assert np.array_equal(di["num_poly_vertices"][sel], ref.draw_info["num_poly_vertices"])
    assert np.array_equal(f.poly[a:a + ref.poly.shape[0]].view(np.uint32), ref.poly.view(np.uint32))
    pset.close()


def vgr_subset(ps, p0, p1):
    vgr = importlib.import_module("vg-renderer_amd")
    c0, c1 = int(ps.path_cmd_begin[p0]), int(ps.path_cmd_begin[p1])
    a0, a1 = int(ps.cmd_arg_off[c0]), int(ps.cmd_arg_off[c1])
    return vgr.PathSetArrays(ps.cmd_type[c0:c1], ps.cmd_arg_off[c0:c1 + 1] - a0, ps.args[a0:a1], ps.path_cmd_begin[p0:p1 + 1] - c0)


def test_full_size_round_joins_polylines(rt, gpu_ctx, wl, oracle):
    """BASELINE config 4 at full size: 10 000 random-walk polylines x 1000 segments, strokeAA width 6, Round caps and
    Round joins (the only data-dependent mesh sizes: k_mesh_prepare counts them). ~80 M vertices / 360 M indices.
      - mesh table is a consistent exclusive scan, one stroke mesh per polyline, every mesh below 65 536 vertices,
      - every index addresses a vertex of its own mesh,
      - colours are the stroke colour on the core rails and colour & 0x00FFFFFF on the fringe rails,
      - sampled polylines match the oracle bit-exactly (positions, colours, indices)."""
    import torch
    n = 10000
    ps, d = wl.random_walk_polylines(n=n, nseg=1000, seed=5678)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    r = rt.tessellate(gpu_ctx, pset, dd, n, to_host=False)
    assert r.sizes["num_meshes"] == n and r.sizes["num_serial_draws"] == 0
    meshes = r.bufs.meshes[:n * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    nvm = meshes["num_vertices"].astype(np.int64)
    nim = meshes["num_indices"].astype(np.int64)
    assert int(nvm.sum()) == r.sizes["num_vertices"] and int(nim.sum()) == r.sizes["num_indices"]
    assert int(nvm.max()) < 65536 and 70_000_000 < r.sizes["num_vertices"] < 90_000_000
    assert np.array_equal(meshes["first_vertex"][1:], np.cumsum(nvm.astype(np.uint64))[:-1])
    assert np.array_equal(meshes["first_index"][1:], np.cumsum(nim.astype(np.uint64))[:-1])
    assert np.array_equal(meshes["draw"], np.arange(n, dtype=np.uint32))
    ni = r.sizes["num_indices"]
    nv = r.sizes["num_vertices"]
    owner = torch.repeat_interleave(torch.arange(n, device=r.bufs.idx.device), torch.from_numpy(nim).to(r.bufs.idx.device))
    lim = torch.from_numpy(nvm).to(owner.device)[owner]
    idx = r.bufs.idx[:ni].view(torch.int16).to(torch.int32) & 0xFFFF
    assert bool((idx < lim).all().item())
    del owner, lim, idx
    col = r.bufs.color[:nv].view(torch.int32)
    c = int(d["stroke_color"][0])
    c_signed = c - (1 << 32) if c >= (1 << 31) else c
    assert bool(((col == c_signed) | (col == (c & 0x00FFFFFF))).all().item())
    rs = np.random.RandomState(4)
    for p in [0, n - 1] + [int(x) for x in rs.randint(1, n - 1, size=6)]:
        sub = vgr_subset(ps, p, p + 1)
        dsub = d[p:p + 1].copy()
        dsub["path"] = 0
        ref = oracle.tessellate(sub, dsub)
        fv, fi = int(meshes["first_vertex"][p]), int(meshes["first_index"][p])
        assert int(nvm[p]) == ref.sizes["num_vertices"] and int(nim[p]) == ref.sizes["num_indices"], p
        assert np.array_equal(r.bufs.pos[fv:fv + int(nvm[p])].cpu().numpy().view(np.uint32), ref.pos.view(np.uint32)), p
        assert np.array_equal(r.bufs.idx[fi:fi + int(nim[p])].cpu().numpy().view(np.uint16), ref.idx), p
        assert np.array_equal(r.bufs.color[fv:fv + int(nvm[p])].cpu().numpy().view(np.uint32), ref.color), p
    # the steady-state entry (single-pass flatten into the heap) must produce the same streams byte for byte
    bufs2 = rt.MeshBuffers(dd.device, nv, ni, n)
    rt.tessellate_async(gpu_ctx, pset, dd, n, bufs2)
    torch.cuda.synchronize()
    assert int(bufs2.dev_status.item()) == 0
    assert torch.equal(bufs2.pos[:nv].view(torch.int32), r.bufs.pos[:nv].view(torch.int32))
    assert torch.equal(bufs2.idx[:ni], r.bufs.idx[:ni]) and torch.equal(bufs2.color[:nv], r.bufs.color[:nv])
    assert torch.equal(bufs2.meshes[:n * 32], r.bufs.meshes[:n * 32])
    del r, col, bufs2
    torch.cuda.empty_cache()
    pset.close()


def test_full_size_tiger_assembly(rt, gpu_ctx, wl):
    """Draw-command assembly at full size (Tiger x10k, 65 536-vertex buffers, ~6 300 draw commands): size-independent
    properties of the greedy partition and of the rebased index stream.
      - the commands tile the vertex / index / mesh streams in order, each holds <= 65 536 vertices,
      - greedy rule (vg.cpp:5327): the first mesh of command k+1 would not have fitted into command k,
      - rebased index - plain index == vertices in front of the mesh inside its vertex buffer (mod 2^16), checked on
        three 40 M-index windows of the stream; vertex streams are byte-identical to the unassembled run."""
    import torch
    K = 10000
    ps, ops = wl.tiger_paths()
    draws = wl.tiger_draws(ops, K)
    n = draws.shape[0]
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(draws)
    del draws
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, n)
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    plain = rt.MeshBuffers(dd.device, nv, ni, nm)
    rt.tessellate_async(gpu_ctx, pset, dd, n, plain)
    asm = rt.MeshBuffers(dd.device, nv, ni, nm)
    cap = 2 * (nv // 65536) + 2
    cmds_dev = torch.zeros(cap * 48, dtype=torch.uint8, device=dd.device)
    ncmd = torch.zeros(1, dtype=torch.int64, device=dd.device)
    gpu_ctx.set_assembly(cmds_dev, 0, ncmd)
    try:
        # arming changes what a count sizes (the unarmed count above chose the template mode, which does not assemble): count again
        assert rt.tessellate_count(gpu_ctx, pset, dd, n) == sizes
        rt.tessellate_async(gpu_ctx, pset, dd, n, asm)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_assembly(None)
    assert int(plain.dev_status.item()) == 0 and int(asm.dev_status.item()) == 0
    T = int(ncmd.item())
    assert int(asm.dev_sizes[9].item()) == T and nv // 65536 <= T <= cap
    c = cmds_dev[:T * 48].cpu().numpy().view(rt.capi.drawcmd_dtype)
    m = asm.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    assert np.array_equal(c["vertex_buffer"], np.arange(T, dtype=np.uint32))
    assert int(c["num_vertices"].max()) <= 65536
    assert int(c["first_vertex"][0]) == 0 and np.array_equal(c["first_vertex"][1:], np.cumsum(c["num_vertices"].astype(np.uint64))[:-1])
    assert int(c["first_index"][0]) == 0 and np.array_equal(c["first_index"][1:], np.cumsum(c["num_indices"].astype(np.uint64))[:-1])
    assert int(c["first_mesh"][0]) == 0 and np.array_equal(c["first_mesh"][1:], np.cumsum(c["num_meshes"].astype(np.uint64))[:-1])
    assert int(c["num_vertices"].astype(np.uint64).sum()) == nv and int(c["num_meshes"].astype(np.uint64).sum()) == nm
    assert np.array_equal(c["first_vertex"], m["first_vertex"][c["first_mesh"].astype(np.int64)])
    nxt = m["num_vertices"][c["first_mesh"][1:].astype(np.int64)].astype(np.int64)
    assert bool((c["num_vertices"][:-1].astype(np.int64) + nxt > 65536).all())
    assert torch.equal(asm.pos.view(torch.int32), plain.pos.view(torch.int32)) and torch.equal(asm.color, plain.color)
    assert torch.equal(asm.meshes, plain.meshes)
    # per-mesh base = first_vertex(mesh) - first_vertex(first mesh of its command)
    cmd_of_mesh = np.repeat(np.arange(T), c["num_meshes"].astype(np.int64))
    base = (m["first_vertex"] - c["first_vertex"][cmd_of_mesh]).astype(np.int64)
    assert int(base.max()) < 65536
    fi = m["first_index"].astype(np.int64)
    for lo_mesh in (0, nm // 2, nm - 120000):
        hi_mesh = min(nm, lo_mesh + 100000)
        a, b = int(fi[lo_mesh]), int(fi[hi_mesh]) if hi_mesh < nm else ni
        cnt = torch.from_numpy(m["num_indices"][lo_mesh:hi_mesh].astype(np.int64)).to(dd.device)
        per_index_base = torch.repeat_interleave(torch.from_numpy(base[lo_mesh:hi_mesh]).to(dd.device), cnt)
        diff = (asm.idx[a:b].to(torch.int64) - plain.idx[a:b].to(torch.int64)) & 0xFFFF
        assert torch.equal(diff, per_index_base & 0xFFFF), lo_mesh
        del per_index_base, diff
    del plain, asm
    torch.cuda.empty_cache()
    pset.close()


def test_full_size_tiger_through_shape_cache(rt, gpu_ctx, wl, oracle):
    """Tiger x10k through the shape cache (one drawing tessellated once, 10 000 submissions): totals, mesh-table scan,
    every instance's colour / index block identical to the cache, sampled instances bit-exact against the oracle."""
    import torch
    K = 10000
    ps, d1 = wl.tiger(1)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d1)
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, d1.shape[0])
    cb = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    rt.tessellate_emit(gpu_ctx, pset, dd, d1.shape[0], cb)
    cache = rt.MeshCache(gpu_ctx, cb, sizes, dd, d1.shape[0])
    ref_cache = oracle.cache_localize(d1, oracle.tessellate(ps, d1))
    inst = np.zeros(K, dtype=rt.capi.cache_instance_dtype)
    inst["num_meshes"] = cache.nm
    rs = np.random.RandomState(9)
    ang = rs.uniform(0, 6.28, K).astype(np.float32)
    inst["mtx"][:, 0] = np.cos(ang); inst["mtx"][:, 1] = np.sin(ang); inst["mtx"][:, 2] = -np.sin(ang); inst["mtx"][:, 3] = np.cos(ang)
    inst["mtx"][:, 4] = 37.0 * (np.arange(K) % 100)
    inst["mtx"][:, 5] = 41.0 * (np.arange(K) // 100)
    raw = torch.from_numpy(inst.view(np.uint8).reshape(-1).copy()).to(dd.device)
    nv1, ni1, nm1 = cache.nv, cache.ni, cache.nm
    out = rt.MeshBuffers(dd.device, nv1 * K, ni1 * K, nm1 * K)
    rt.cache_submit(gpu_ctx, cache, raw, K, out)
    torch.cuda.synchronize()
    assert int(out.dev_status.item()) == 0
    sz = out.dev_sizes.cpu().numpy()
    assert (int(sz[3]), int(sz[4]), int(sz[2])) == (nv1 * K, ni1 * K, nm1 * K)
    assert bool((out.color[:nv1 * K].view(K, nv1) == cb.color[:nv1].view(1, nv1)).all().item())
    assert bool((out.idx[:ni1 * K].view(K, ni1) == cb.idx[:ni1].view(1, ni1)).all().item())
    m = out.meshes[:nm1 * K * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    assert np.array_equal(m["first_vertex"][1:], np.cumsum(m["num_vertices"].astype(np.uint64))[:-1])
    assert np.array_equal(m["first_index"][1:], np.cumsum(m["num_indices"].astype(np.uint64))[:-1])
    assert np.array_equal(m["draw"], np.repeat(np.arange(K, dtype=np.uint32), nm1))
    for i in [0, K - 1] + [int(x) for x in rs.randint(1, K - 1, size=6)]:
        ref = oracle.cache_submit(ref_cache, inst[i:i + 1])
        got = out.pos[i * nv1:(i + 1) * nv1].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), ref.pos.view(np.uint32)), i
    del out
    torch.cuda.empty_cache()
    pset.close()
