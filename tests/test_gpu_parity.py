"""GPU parity: HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.
Bar: indices / colours / mesh tables / sub-path tables bit-exact, positions bit-exact (0 ulp; the
north-star tolerance is 1e-4)."""
import importlib

import numpy as np
import pytest

from util import assert_flat_equal, assert_mesh_equal, run_async

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    return importlib.import_module("vg-renderer_amd.runtime")


def _run_flat(rt, ctx, ps, draws, xform):
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(draws)
    r = rt.flatten(ctx, pset, dd, draws.shape[0], apply_transform=xform)
    pset.close()
    return r


def _run_mesh(rt, ctx, ps, draws):
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(draws)
    r = rt.tessellate(ctx, pset, dd, draws.shape[0])
    pset.close()
    return r


def test_config0_single_cubic(rt, gpu_ctx, wl, oracle):
    ps, d = wl.single_cubic()
    got = _run_mesh(rt, gpu_ctx, ps, d)
    ref = oracle.tessellate(ps, d)
    assert got.sizes["num_poly_vertices"] == 17
    assert (got.sizes["num_vertices"], got.sizes["num_indices"]) == (68, 300)
    assert_mesh_equal(got, ref, "config0")


@pytest.mark.parametrize("box", [10.0, 100.0, 1000.0])
def test_flatten_random_cubics(rt, gpu_ctx, wl, oracle, box):
    ps, d = wl.random_cubics(20000, seed=1234, box=box)
    got = _run_flat(rt, gpu_ctx, ps, d, False)
    ref = oracle.flatten(ps, d, apply_transform=False)
    assert_flat_equal(got, ref, "cubics box=%g" % box)


@pytest.mark.parametrize("seed", list(range(12)))
def test_flatten_fuzz_all_commands(rt, gpu_ctx, wl, oracle, seed):
    ps = wl.fuzz_paths(seed, npaths=96)
    d = wl.fuzz_draws(ps, seed)
    for xform in (False, True):
        got = _run_flat(rt, gpu_ctx, ps, d, xform)
        ref = oracle.flatten(ps, d, apply_transform=xform)
        assert_flat_equal(got, ref, "fuzz seed=%d xform=%s" % (seed, xform))


@pytest.mark.parametrize("seed", list(range(12)))
def test_tessellate_fuzz_all_strokers(rt, gpu_ctx, wl, oracle, seed):
    ps = wl.fuzz_paths(100 + seed, npaths=96)
    d = wl.fuzz_draws(ps, 100 + seed)
    got = _run_mesh(rt, gpu_ctx, ps, d)
    ref = oracle.tessellate(ps, d)
    assert_mesh_equal(got, ref, "fuzz seed=%d" % seed)


def test_tiger_small(rt, gpu_ctx, wl, oracle):
    ps, d = wl.tiger(3)
    got = _run_mesh(rt, gpu_ctx, ps, d)
    ref = oracle.tessellate(ps, d)
    assert_mesh_equal(got, ref, "tiger x3")


@pytest.mark.parametrize("cap,join", [(0, 0), (1, 1), (2, 2), (1, 0), (0, 1)])
def test_long_polylines(rt, gpu_ctx, wl, oracle, cap, join):
    ps, d = wl.random_walk_polylines(n=40, nseg=1000, seed=5678, cap=cap, join=join)
    got = _run_mesh(rt, gpu_ctx, ps, d)
    ref = oracle.tessellate(ps, d)
    assert_mesh_equal(got, ref, "polylines cap=%d join=%d" % (cap, join))


def test_async_entry_point_matches_two_phase(rt, gpu_ctx, wl):
    import torch
    ps, d = wl.tiger(2)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    a = rt.tessellate(gpu_ctx, pset, dd, d.shape[0])
    bufs = rt.MeshBuffers(dd.device, a.sizes["num_vertices"], a.sizes["num_indices"], a.sizes["num_meshes"])
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    sz = bufs.dev_sizes.cpu().numpy()
    assert int(sz[3]) == a.sizes["num_vertices"] and int(sz[4]) == a.sizes["num_indices"]
    nv, ni = a.sizes["num_vertices"], a.sizes["num_indices"]
    assert np.array_equal(bufs.idx[:ni].cpu().numpy().view(np.uint16), a.idx)
    assert np.array_equal(bufs.pos[:nv].cpu().numpy().view(np.uint32), a.pos.view(np.uint32))
    # too-small output buffers are reported, not overrun
    small = rt.MeshBuffers(dd.device, nv // 2, ni // 2, a.sizes["num_meshes"])
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], small)
    torch.cuda.synchronize()
    assert int(small.dev_status.item()) == 4  # VGX_E_NOSPACE
    pset.close()


@pytest.mark.parametrize("seed", [200, 201, 202])
def test_stroker_level_entry(rt, gpu_ctx, wl, oracle, seed):
    """vgx_stroke_*: the strokerXXX-level boundary (vertex lists in, meshes out). Feed it the oracle's own
    transformed polylines and compare every mesh with the oracle's mesh for the same (draw, sub-path, kind)."""
    import torch
    ps = wl.fuzz_paths(seed, npaths=64)
    d = wl.fuzz_draws(ps, seed)
    ref = oracle.tessellate(ps, d, want_flat=True)
    nsubs = ref.subpaths.shape[0]
    sub_draw = np.repeat(np.arange(d.shape[0], dtype=np.int32), ref.draw_info["num_subpaths"])
    poly = torch.from_numpy(ref.poly.copy()).cuda()
    subs = torch.from_numpy(ref.subpaths.view(np.uint8).copy()).cuda()
    sd = torch.from_numpy(sub_draw).cuda()
    dd = rt.upload_draws(d)
    got = rt.stroke(gpu_ctx, poly, subs, sd, nsubs, dd, d.shape[0])
    assert got.sizes["num_meshes"] == ref.sizes["num_meshes"]
    assert got.sizes["num_vertices"] == ref.sizes["num_vertices"] and got.sizes["num_indices"] == ref.sizes["num_indices"]
    sub0 = ref.draw_info["first_subpath"]
    key_ref = {}
    for m in ref.meshes:
        dr = int(m["draw"])
        gsub = int(sub0[dr]) + (int(m["subpath_kind"]) & 0x0FFFFFFF)
        key_ref[(gsub, int(m["subpath_kind"]) >> 28)] = m
    for m in got.meshes:
        r = key_ref[(int(m["subpath_kind"]) & 0x0FFFFFFF, int(m["subpath_kind"]) >> 28)]
        assert int(m["num_vertices"]) == int(r["num_vertices"]) and int(m["num_indices"]) == int(r["num_indices"])
        gv, gi, rv, ri = int(m["first_vertex"]), int(m["first_index"]), int(r["first_vertex"]), int(r["first_index"])
        nv, ni = int(m["num_vertices"]), int(m["num_indices"])
        assert np.array_equal(got.idx[gi:gi + ni], ref.idx[ri:ri + ni])
        assert np.array_equal(got.color[gv:gv + nv], ref.color[rv:rv + nv])
        assert np.array_equal(got.pos[gv:gv + nv].view(np.uint32), ref.pos[rv:rv + nv].view(np.uint32))


@pytest.mark.parametrize("seed", [300, 301, 302, 303, 304, 305])
def test_async_single_pass_fuzz(rt, gpu_ctx, wl, oracle, seed):
    """vgx_tessellate (steady-state entry: single-pass flatten into the polyline heap, no host round trip) on fuzz
    batches with every command / stroker kind, against the oracle."""
    import torch
    ps = wl.fuzz_paths(seed, npaths=96)
    d = wl.fuzz_draws(ps, seed)
    d = np.concatenate([d, d[::-1], d])  # several instances, different order
    ref = oracle.tessellate(ps, d)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    assert sizes["num_vertices"] == ref.sizes["num_vertices"]
    bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    bufs.pos.fill_(float("nan"))
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]

    class G:
        pass
    got = G()
    got.sizes = sizes
    got.pos = bufs.pos[:nv].cpu().numpy()
    got.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
    got.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
    got.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    assert_mesh_equal(got, ref, "async fuzz seed=%d" % seed)
    pset.close()


def test_async_steady_state_with_changed_draw_parameters(rt, gpu_ctx, wl, oracle):
    """The steady-state entry point must not depend on anything the sizing pass (vgx_tessellate_count) left in the scratch:
    after ONE count on a batch, the draws are rewritten in place -- other colours, stroke widths, fringe, a mirrored
    transform (flips the fill orientation sign the per-mesh constants carry, stroker.cpp:721-723), other caps / joins of
    the same size class -- and vgx_tessellate alone must reproduce the oracle for the NEW parameters. (Regression: the
    per-mesh constants are written by the single-pass flatten stage itself, not by a k_mesh_prepare pass of the count.)"""
    import torch
    ps, d = wl.tiger(3)
    rs = np.random.RandomState(5)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    d2 = d.copy()
    d2["fill_color"] = rs.randint(0, 2 ** 32, size=d.shape[0], dtype=np.uint64).astype(np.uint32)
    d2["stroke_color"] = rs.randint(0, 2 ** 32, size=d.shape[0], dtype=np.uint64).astype(np.uint32)
    stroked = (d2["stroke_flags"] & rt.capi.STROKE_ENABLE) != 0
    d2["stroke_width"][stroked] = rs.uniform(1.5, 6.0, size=int(stroked.sum())).astype(np.float32)
    d2["fringe"] = np.float32(0.75)
    d2["mtx"][:, 0] = -1.0  # mirror in x: every polygon changes orientation
    d2["mtx"][:, 4] = 900.0 - d2["mtx"][:, 4]
    ref = oracle.tessellate(ps, d2)
    assert ref.sizes["num_vertices"] == sizes["num_vertices"] and ref.sizes["num_indices"] == sizes["num_indices"]
    dd.copy_(torch.from_numpy(d2.view(np.uint8).reshape(-1).copy()))
    bufs.pos.fill_(float("nan"))
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]

    class G:
        pass
    got = G()
    got.sizes = sizes
    got.pos = bufs.pos[:nv].cpu().numpy()
    got.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
    got.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
    got.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    assert_mesh_equal(got, ref, "steady state, changed draws")
    # and a different batch of the same capacity: the draws in another order (other offsets everywhere)
    d3 = d2[rs.permutation(d2.shape[0])]
    ref3 = oracle.tessellate(ps, d3)
    dd.copy_(torch.from_numpy(d3.view(np.uint8).reshape(-1).copy()))
    bufs.pos.fill_(float("nan"))
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    got.pos = bufs.pos[:nv].cpu().numpy()
    got.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
    got.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
    got.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    assert_mesh_equal(got, ref3, "steady state, permuted draws")
    pset.close()


def test_async_single_pass_long_paths(rt, gpu_ctx, wl, oracle):
    """Draws larger than a heap block (exactly sized regions) and many-chunk segments."""
    import torch
    ps, d = wl.random_walk_polylines(n=12, nseg=5000, seed=99, cap=1, join=1)
    ref = oracle.tessellate(ps, d)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    nv, ni = sizes["num_vertices"], sizes["num_indices"]
    assert np.array_equal(bufs.idx[:ni].cpu().numpy().view(np.uint16), ref.idx)
    assert np.array_equal(bufs.pos[:nv].cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
    pset.close()


def _async_result(rt, gpu_ctx, ps, d):
    import torch
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    bufs.pos.fill_(float("nan"))
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    status = int(bufs.dev_status.item())
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]

    class G:
        pass
    got = G()
    got.sizes = sizes
    got.status = status
    got.pos = bufs.pos[:nv].cpu().numpy()
    got.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
    got.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
    got.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    pset.close()
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 5, 8])
def test_async_large_batch_launch_sequence_on_small_batches(rt, wl, oracle, seed, monkeypatch):
    """Batches of at most VGX_SMALL_DRAWS draws take the frame-sized path of vgx_tessellate (three one-workgroup kernels
    around k_flatten_build). VGX_NO_SMALL=1 (read at vgx_create) sends the same fuzz batches -- every command, degenerate
    draws, serial shapes, Round joins -- through the large-batch launch sequence, which the full-size tests only exercise
    with the Tiger drawing."""
    monkeypatch.setenv("VGX_NO_SMALL", "1")
    ctx = rt.Context(0)
    ps = wl.fuzz_paths(seed, npaths=96)
    d = wl.fuzz_draws(ps, seed)
    d = np.concatenate([d, d[::-1]])
    ref = oracle.tessellate(ps, d)
    got = _async_result(rt, ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "large-batch sequence, fuzz seed=%d" % seed)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("small", [True, False])
def test_async_interleaved_mesh_classes_force_the_search_paths(rt, wl, vgr, oracle, small, monkeypatch):
    """k_fill and k_stroke find a lane's mesh in a 64-entry window of the mesh table. The fill element stream has a
    zero-length entry for every stroke mesh (and vice versa), so a run of more than 63 stroke-only draws between two filled
    ones (resp. fill-only draws between two stroked ones) puts more than 63 records inside ONE 64-element chunk: the window
    cannot cover it and every lane searches its mesh in memory (fill_chunk_slow / the search branch of k_stroke)."""
    if not small:
        monkeypatch.setenv("VGX_NO_SMALL", "1")
    ctx = rt.Context(0)
    b = vgr.PathSetBuilder()
    b.begin_path(); b.move_to(0, 0); b.line_to(30, 0); b.line_to(15, 25); b.close(); b.end_path()          # 0: triangle
    b.begin_path(); b.move_to(0, 0); b.line_to(40, 10); b.end_path()                                         # 1: a line
    b.begin_path(); b.move_to(0, 0); b.cubic_to(10, 30, 40, 30, 50, 0); b.line_to(25, -20); b.close(); b.end_path()  # 2: blob
    ps = b.arrays()
    pm = importlib.import_module("vg-renderer_amd.pathset")
    rs = np.random.RandomState(9)
    kinds = []  # (path, fill, stroke)
    for rep in range(6):
        kinds += [(2, True, False)] * 2 + [(1, False, True)] * int(rs.randint(64, 150)) + [(0, True, True)]
        kinds += [(0, True, False)] * int(rs.randint(64, 150)) + [(1, False, True)] * 3
    d = pm.make_draws(len(kinds))
    for i, (p, f, st) in enumerate(kinds):
        d["path"][i] = p
        if f:
            wl.set_fill(d, i, 0xFF00A0FF, aa=bool(i % 3))
        if st:
            wl.set_stroke(d, i, 0xFF2080FF, 1.5 + (i % 4), rt.capi.CAP_BUTT + i % 3, rt.capi.JOIN_MITER + i % 3, aa=bool(i % 2))
    d["mtx"][:, 4] = rs.uniform(0, 500, len(kinds)).astype(np.float32)
    d["mtx"][:, 5] = rs.uniform(0, 500, len(kinds)).astype(np.float32)
    ref = oracle.tessellate(ps, d)
    got = _async_result(rt, ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "interleaved mesh classes (small=%s)" % small)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("waves", ["3", "17"])
def test_async_heap_block_switches(rt, wl, oracle, waves, monkeypatch):
    """The multi-kernel pipeline's single-pass flatten (k_flatten_build: what vgx_tessellate runs for batches that are not
    instanced) with only a few waves: every wave fills many 8192-vertex heap blocks, so chunks that do not fit, block
    switches and the move of the sub-path that spans the switch all happen in a batch the oracle can check completely
    (at the default 4096 waves that needs > 33 M polyline vertices). Options are read at vgx_create: own context."""
    monkeypatch.setenv("VGX_BUILD_WAVES", waves)
    gpu_ctx = rt.Context(0)
    ps, d = wl.tiger(24)
    ref = oracle.tessellate(ps, d)
    got = _async_result(rt, gpu_ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "tiger x24, %s build waves" % waves)
    ps = wl.fuzz_paths(411, npaths=128)
    d = wl.fuzz_draws(ps, 411)
    d = np.concatenate([d] * 12)
    ref = oracle.tessellate(ps, d)
    got = _async_result(rt, gpu_ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "fuzz x12, %s build waves" % waves)
    gpu_ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("waves", ["2", None])
def test_async_very_long_subpaths(rt, wl, oracle, waves, monkeypatch):
    """Sub-paths of 30 001 vertices built from single LINE_TO commands (470 chunks each). Multi-kernel pipeline: they
    outgrow several heap blocks, are moved with geometric growth, and their total feeds the heap sizing
    (long_subpath_vertices)."""
    if waves:
        monkeypatch.setenv("VGX_BUILD_WAVES", waves)
    gpu_ctx = rt.Context(0)
    ps, d = wl.random_walk_polylines(n=5, nseg=30000, seed=7, cap=0, join=0, width=3.0)
    d["stroke_flags"] &= ~np.uint32(rt.capi.STROKE_AA)  # 2 rails: 60 002 vertices per mesh stay below 65 536
    ref = oracle.tessellate(ps, d)
    got = _async_result(rt, gpu_ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "5 x 30001-vertex polylines")
    gpu_ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["tiger", "fuzz", "bigcubics"])
def test_async_pooled_walk(rt, wl, oracle, workload, monkeypatch):
    """VGX_WALK=pool: the wave subdivides all cubics of a chunk together (task LIFO in LDS, ballot + popcount compaction,
    leaf ranks from per-command bit masks) instead of one cubic per lane; deeper cubics fall back to the per-lane walk."""
    monkeypatch.setenv("VGX_WALK", "pool")
    ctx = rt.Context(0)
    if workload == "tiger":
        ps, d = wl.tiger(24)
    elif workload == "fuzz":
        ps = wl.fuzz_paths(512, npaths=200, with_shapes=False, degenerate=False)
        d = wl.fuzz_draws(ps, 512, ndraws=1500)
    else:
        ps, d = wl.random_cubics(3000, seed=78, box=1000.0)
        wl.set_fill(d, slice(None), 0xFF336699, aa=True)
        wl.set_stroke(d, slice(None), 0xFF2080FF, 2.0, 0, 0, aa=True)
    ref = oracle.tessellate(ps, d)
    got = _async_result(rt, ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "pooled walk, %s" % workload)
    ctx.close()


@pytest.mark.gpu
def test_async_call_is_graph_capturable(rt, gpu_ctx, wl, oracle):
    """Steady state: vgx_tessellate only enqueues kernels / memsets on the caller's stream (no allocation, no host
    sync), so a frame can be captured into a HIP graph once and replayed; the replay must reproduce the oracle."""
    import torch
    ps, d = wl.tiger(3)
    ref = oracle.tessellate(ps, d)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)  # warm: scratch is sized, nothing left to allocate
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
    for _ in range(3):
        bufs.pos.fill_(float("nan"))
        bufs.idx.zero_()
        bufs.dev_status.fill_(77)
        g.replay()
        torch.cuda.synchronize()
        assert int(bufs.dev_status.item()) == 0
        assert np.array_equal(bufs.pos[:nv].cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
        assert np.array_equal(bufs.idx[:ni].cpu().numpy().view(np.uint16), ref.idx)
        assert np.array_equal(bufs.color[:nv].cpu().numpy().view(np.uint32), ref.color)
    del g
    pset.close()


@pytest.mark.gpu
@pytest.mark.parametrize("field,value", [("scale", 0.0), ("scale", float("nan")), ("tess_tol", 0.0), ("tess_tol", -1.0), ("tess_tol", 1e-30),
                                          ("fringe", float("inf")), ("stroke_width", float("nan")), ("mtx", float("nan"))])
def test_hostile_draw_records_are_rejected_not_subdivided(rt, gpu_ctx, wl, field, value):
    """Draw records live in device memory the host never reads. Parameters that would drive the adaptive subdivision
    to the limits of float (zero / NaN tolerance or scale ...) must come back as a status, quickly, not as a hang."""
    import torch
    ps, d = wl.tiger(1)
    d = d.copy()
    if field == "mtx":
        d["mtx"][5, 2] = value
    else:
        d[field][7] = value
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    with pytest.raises(rt.VgxError) as ei:
        rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    assert ei.value.status == rt.capi.VGX_E_NONFINITE
    good = wl.tiger(1)[1]
    sizes = rt.tessellate_count(gpu_ctx, pset, rt.upload_draws(good), good.shape[0])
    bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)  # the asynchronous entry reports through the status word
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == rt.capi.VGX_E_NONFINITE
    pset.close()


@pytest.mark.gpu
def test_unbounded_shapes_end_as_status_not_as_a_hang(rt, gpu_ctx, wl):
    """A circle / rounded rect / Round cap whose radius is beyond ~4e6 tolerances makes the reference compute
    ceil(pi / acos(1)) = ceil(pi / 0) points and cast the infinity to uint32 (path.cpp:307,602; stroker.cpp:1013-1014).
    The device saturates the count above what a mesh may hold, so the batch comes back as VGX_E_MESH_TOO_LARGE after
    bounded work; arc angles that would keep pathArc's wrap loops spinning are rejected when the path set is built."""
    import torch
    pm = importlib.import_module("vg-renderer_amd.pathset")
    b = pm.PathSetBuilder()
    b.begin_path(); b.circle(0.0, 0.0, 1.0e9); b.end_path()
    b.begin_path(); b.rounded_rect(0.0, 0.0, 4.0e9, 4.0e9, 1.0e9); b.end_path()
    b.begin_path(); b.move_to(0, 0); b.line_to(10, 0); b.end_path()
    ps = b.arrays()
    d = pm.make_draws(3)
    d["path"] = [0, 1, 2]
    wl.set_fill(d, slice(0, 2), 0xFF0000FF, aa=True)
    wl.set_stroke(d, slice(2, 3), 0xFF00FF00, 8.0e9, rt.capi.CAP_ROUND, rt.capi.JOIN_ROUND, aa=True)
    d["stroke_width"][2] = 8.0e9  # (set_stroke clamps like vg.cpp:3416; force the raw width the stroker would see)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    with pytest.raises(rt.VgxError) as ei:
        rt.tessellate_count(gpu_ctx, pset, dd, 3)
    assert ei.value.status == rt.capi.VGX_E_MESH_TOO_LARGE
    pset.close()
    b = pm.PathSetBuilder()
    b.begin_path(); b.arc(0.0, 0.0, 10.0, 1.0e9, 0.0, True); b.end_path()
    with pytest.raises(rt.VgxError) as ei:
        rt.PathSet(gpu_ctx, b.arrays())
    assert ei.value.status == rt.capi.VGX_E_INVALID_ARG


@pytest.mark.parametrize("seed", [300, 301, 302])
def test_sse_index_order_option(rt, gpu_ctx, wl, oracle, seed):
    """VGX_FILL_INDEX_ORDER_SSE on every AA fill: the index stream is the one the reference's SSE2 strokerConvexFillAA writes
    (stroker.cpp:610-701; the oracle's restated order is pinned against that build in tests/test_oracle_golden.py), everything
    else is the scalar build's -- through the two-phase entry, the asynchronous one, the instanced kernel (tiger instances) and
    with draw-command assembly armed (the index base is added to the reordered values as well)."""
    import torch
    flag = np.uint32(rt.capi.FILL_INDEX_ORDER_SSE)
    if seed == 300:
        ps, d = wl.tiger(40)       # > 2048 draws: k_flatten_inst + the large-batch launch sequence
    else:
        ps = wl.fuzz_paths(seed, npaths=96)
        d = wl.fuzz_draws(ps, seed)
    d = d.copy()
    half = np.arange(d.shape[0]) % 2 == 0
    d["fill_flags"][half] |= flag  # mixed: meshes with and without the option in one batch
    ref = oracle.tessellate(ps, d)
    assert not np.array_equal(ref.idx, oracle.tessellate(ps, wl_without(d, flag)).idx)
    got = _run_mesh(rt, gpu_ctx, ps, d)
    assert_mesh_equal(got, ref, "sse order, two-phase, seed=%d" % seed)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    assert np.array_equal(bufs.idx[:ni].cpu().numpy().view(np.uint16), ref.idx)
    assert np.array_equal(bufs.pos[:nv].cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
    # assembly armed: the rebased index buffer of the oracle's assembler over the reordered meshes
    max_vb = int(max(700, ref.meshes["num_vertices"].max()))  # a mesh larger than a vertex buffer is an error of its own
    st, rcmds, ridx = oracle.assemble(ref.meshes, ref.idx, max_vb)
    assert st == 0
    cmds = torch.zeros((2 * (nv // max_vb) + 2) * 48, dtype=torch.uint8, device=dd.device)
    ncmd = torch.zeros(1, dtype=torch.int64, device=dd.device)
    gpu_ctx.set_assembly(cmds, max_vb, ncmd)
    try:
        rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_assembly(None)
    assert int(ncmd.item()) == len(rcmds)
    assert np.array_equal(bufs.idx[:ni].cpu().numpy().view(np.uint16), ridx)
    pset.close()


def wl_without(d, flag):
    e = d.copy()
    e["fill_flags"] &= ~flag
    return e


@pytest.mark.gpu
@pytest.mark.parametrize("seed,npaths,degenerate", [(0, 2600, False), (1, 2600, True), (2, 5000, True), (3, 150, False), (4, 150, True), (5, 40, True)])
def test_async_thin_path_sets_static_layout(rt, wl, oracle, seed, npaths, degenerate, monkeypatch):
    """Path sets of moveTo / lineTo / close paths only take k_flatten_thin in vgx_tessellate (vgx_thin.h: the polyline layout of
    such a path is decided when the set is created, the kernel is gather - transform - scatter): sub-paths of one and two
    vertices, polygons closing onto their first point (the popped vertex), runs longer than a chunk, draws of degenerate paths
    through the exact builder -- every path drawn once (no instancing), large-batch and frame-sized launch sequences, against
    the oracle; and the same batch through k_flatten_build (VGX_THIN_STATIC=0) for the other side of the switch."""
    ps = wl.thin_fuzz_paths(seed, npaths=npaths, degenerate=degenerate)
    d = wl.fuzz_draws(ps, seed)
    d = d[np.random.RandomState(seed).permutation(d.shape[0])]
    ref = oracle.tessellate(ps, d)
    import os
    on = os.environ.get("VGX_THIN_STATIC", "1")  # ("2": the kernel instance with two command instances per thread)
    for static in (on if on != "0" else "1", "0"):
        monkeypatch.setenv("VGX_THIN_STATIC", static)
        ctx = rt.Context(0)
        got = _async_result(rt, ctx, ps, d)
        assert got.status == 0
        assert got.sizes["num_vertices"] == ref.sizes["num_vertices"]
        assert_mesh_equal(got, ref, "thin set seed=%d static=%s" % (seed, static))
        ctx.close()


@pytest.mark.gpu
def test_async_thin_path_set_steady_state_other_draws(rt, gpu_ctx, wl, oracle):
    """k_flatten_thin in the steady state: counted with one draw list, called with another order, other transforms and other
    fill / stroke switches (the draw's mesh counts come from the path's table and the draw's flags)."""
    import torch
    ps = wl.thin_fuzz_paths(11, npaths=3000, degenerate=True)
    d = wl.fuzz_draws(ps, 11)
    d2 = wl.fuzz_draws(ps, 12)[::-1].copy()
    ref = oracle.tessellate(ps, d2)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d); dd2 = rt.upload_draws(d2)
    rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    assert gpu_ctx.failure_info()["segment_items"] == 6  # k_flatten_thin builds batches like this one
    nv, ni, nm = ref.sizes["num_vertices"], ref.sizes["num_indices"], ref.sizes["num_meshes"]
    bufs = rt.MeshBuffers(dd.device, nv + 64, ni + 64, nm + 4)
    rt.tessellate_async(gpu_ctx, pset, dd2, d2.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0

    class G:
        pass
    got = G()
    got.sizes = ref.sizes
    got.pos = bufs.pos[:nv].cpu().numpy()
    got.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
    got.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
    got.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    assert_mesh_equal(got, ref, "thin set, steady state with other draws")
    pset.close()


@pytest.mark.parametrize("style", ["round_round", "round_butt_wide", "bevel_square", "miter_round_nonaa", "closed_round"])
def test_long_polylines_through_the_staged_stroke_kernel(rt, wl, oracle, style):
    """Batches whose stroke meshes are ALL long (>= 128 elements) are emitted by k_stroke_long: a chunk that lies inside one mesh writes its
    colours and indices into LDS and the wave copies them out as dense runs (round 6). Long polylines in every general style -- arcs of one
    to many points (wide strokes: chunks that outgrow the stage fall back to per-lane stores), Bevel joins, Round / Square / Butt caps, non-AA
    strokes, closed polygons with Round joins --, enough of them for the large-batch launch sequence; chunks at a mesh's ends span two meshes
    (per-lane stores there). Bit-exact against the reference."""
    capi = rt.capi
    cap, join, width, aa, sigma = {"round_round": (capi.CAP_ROUND, capi.JOIN_ROUND, 6.0, True, 0.5), "round_butt_wide": (capi.CAP_BUTT, capi.JOIN_ROUND, 60.0, True, 1.2),
                                   "bevel_square": (capi.CAP_SQUARE, capi.JOIN_BEVEL, 4.0, True, 0.7), "miter_round_nonaa": (capi.CAP_ROUND, capi.JOIN_MITER, 5.0, False, 0.4),
                                   "closed_round": (capi.CAP_BUTT, capi.JOIN_ROUND, 8.0, True, 0.6)}[style]
    ps, d = wl.random_walk_polylines(700, 333, seed=77, width=width, cap=cap, join=join, turn_sigma=sigma)
    if not aa:
        d["stroke_flags"] = capi.stroke_flags(cap, join, aa=False)
    if style == "closed_round":  # the same polylines closed: pathClose behind every path
        b = importlib.import_module("vg-renderer_amd").PathSetBuilder()
        pts = ps.args.reshape(700, 334, 2)
        for k in range(700):
            b.begin_path()
            b.move_to(float(pts[k, 0, 0]), float(pts[k, 0, 1]))
            for q in pts[k, 1:]:
                b.line_to(float(q[0]), float(q[1]))
            b.close()
            b.end_path()
        ps = b.arrays()
    ctx = rt.Context(0)
    got = run_async(rt, ctx, ps, d, profile=True)
    ref = oracle.tessellate(ps, d)
    assert ref.pos.shape[0] > (1 << 18), "large enough for the large-batch sequence (k_stroke_long is not launched for frame-sized calls)"
    assert got.status == 0
    assert_mesh_equal(got, ref, "long polylines, " + style)
    ctx.close()
