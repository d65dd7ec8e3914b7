"""GPU: EVERY unit of the single-GPU BASELINE configs compared with the reference at full size -- every instance of
Tiger x10k (configs[2], the headline), every mesh of the 10 000 round-join polylines (configs[3]), every path of the
1 M cubics (configs[1]) -- through per-unit integer digests (tests/hashutil.py) computed on the device for the product's
buffers and by one reference PROCESS per host core (tests/ref_hash_worker.py) for the oracle's. No sampling."""
import importlib
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import hashutil as hu

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def rt():
    return importlib.import_module("vg-renderer_amd.runtime")


def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def _reference_rows(which, units):
    """One worker process per host core over contiguous unit ranges; returns the concatenated rows."""
    procs = _cores()
    per = (units + procs - 1) // procs
    tmp = tempfile.mkdtemp(prefix="vgxhash_")
    jobs = []
    for i in range(procs):
        a = i * per
        n = min(per, units - a)
        if n <= 0:
            break
        out = os.path.join(tmp, "part%03d.npy" % i)
        jobs.append((subprocess.Popen([sys.executable, os.path.join(HERE, "ref_hash_worker.py"), which, str(a), str(n), out]), out))
    parts = []
    for p, out in jobs:
        assert p.wait() == 0, "reference worker failed"
        parts.append(np.load(out))
        os.remove(out)
    os.rmdir(tmp)
    return np.concatenate(parts)


@pytest.mark.parametrize("which", ["tiger", "tigerspec", "tigeropen", "tigerbevel"])
def test_every_instance_of_tiger_x10k_matches_the_reference(rt, wl, which):
    """BASELINE configs[2] (and the SURVEY 8(d) drawing as specified, bench.py's tigerspec10k) at full size through the entry
    point bench.py times; digests of positions / colours / indices of all 10 000 instances against the reference's."""
    import torch
    K = 10000
    ps, ops = wl.tiger_spec_paths() if which == "tigerspec" else wl.tiger_paths(closed=which != "tigeropen")  # tigeropen: open strokes (k_tmpl_emit_open)
    d = wl.tiger_draws(ops, K, join=2 if which == "tigerbevel" else 0)  # tigerbevel: Bevel joins (the general element body, k_tmpl_emit_general)
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    assert nv % K == 0 and ni % K == 0
    bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    bufs.pos.fill_(float("nan"))
    rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    ref = _reference_rows(which, K)  # [K, 3, 4]
    got = np.stack([hu.digest_uniform_torch(bufs.pos[:nv].view(torch.int32).view(-1), K), hu.digest_uniform_torch(bufs.color[:nv], K),
                    hu.digest_uniform_torch(bufs.idx[:ni], K)], axis=1)
    bad = np.flatnonzero((got != ref).any(axis=(1, 2)))
    assert bad.shape[0] == 0, ("instances that differ from the reference", bad[:10].tolist(), bad.shape[0])
    # the caller's MESH TABLE of the asynchronous (template) entry, every record of every instance (VERDICT r4 "Weak" 1a): instance 0's
    # records equal the oracle's for one instance of the drawing, and instance k's are instance 0's moved by k instances
    import pyoracle
    P = len(ops)
    one = pyoracle.tessellate(ps, d[:P], kind="reference" if pyoracle.available("reference") else None)
    mpi = nm // K
    assert nm % K == 0 and mpi == one.sizes["num_meshes"]
    mt = bufs.meshes[:nm * 32].view(torch.int64).view(K, mpi, 4)  # first_vertex, first_index, (num_vertices | num_indices << 32), (draw | subpath_kind << 32)
    m0 = torch.from_numpy(np.ascontiguousarray(one.meshes).view(np.int64).reshape(mpi, 4).copy()).to(mt.device)
    k = torch.arange(K, dtype=torch.int64, device=mt.device).view(K, 1)
    assert torch.equal(mt[:, :, 0], m0[:, 0].view(1, mpi) + k * (nv // K)), "mesh table: first_vertex"
    assert torch.equal(mt[:, :, 1], m0[:, 1].view(1, mpi) + k * (ni // K)), "mesh table: first_index"
    assert torch.equal(mt[:, :, 2], m0[:, 2].view(1, mpi).expand(K, mpi)), "mesh table: num_vertices / num_indices"
    assert torch.equal(mt[:, :, 3], m0[:, 3].view(1, mpi) + k * P), "mesh table: draw / sub-path / kind"
    pset.close()
    ctx.close()


def test_every_instance_of_tiger_at_seven_scales_matches_the_reference(rt, wl):
    """bench.py's tiger10k_varied at full size (18 template classes: instances of different sizes): digests of positions / colours /
    indices and the sizes of all 10 000 instances against the reference's."""
    import torch
    K = 10000
    ps, ops = wl.tiger_paths()
    P = len(ops)
    d = wl.tiger_varied_draws(ops, K)
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    assert ctx.failure_info()["segment_items"] == 5
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    bufs.pos.fill_(float("nan"))
    rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    ref = _reference_rows("varied", K)  # [K, 12 + 2]
    mt = bufs.meshes[:nm * 32].view(torch.int64).view(-1, 4)
    draw = (mt[:, 3] & 0xFFFFFFFF).contiguous()
    m0 = torch.searchsorted(draw, torch.arange(K + 1, dtype=torch.int64, device=draw.device) * P)  # first mesh of every instance
    fvm = torch.cat([mt[:, 0], torch.tensor([nv], dtype=torch.int64, device=draw.device)])
    fim = torch.cat([mt[:, 1], torch.tensor([ni], dtype=torch.int64, device=draw.device)])
    fv, fi = fvm[m0[:-1]], fim[m0[:-1]]
    cv, ci = fvm[m0[1:]] - fv, fim[m0[1:]] - fi
    assert np.array_equal(cv.cpu().numpy(), ref[:, 12]) and np.array_equal(ci.cpu().numpy(), ref[:, 13])
    got = np.concatenate([hu.digest_ragged_torch(bufs.pos[:nv].view(torch.int32), 2 * fv, 2 * cv), hu.digest_ragged_torch(bufs.color[:nv], fv, cv),
                          hu.digest_ragged_torch(bufs.idx[:ni], fi, ci, is_u16=True)], axis=1)
    bad = np.flatnonzero((got != ref[:, :12]).any(axis=1))
    assert bad.shape[0] == 0, ("instances that differ from the reference", bad[:10].tolist(), bad.shape[0])
    pset.close()
    ctx.close()


def test_every_instance_of_tiger_x10k_with_round_joins_matches_the_reference(rt, wl):
    """bench.py's tiger10k_round at full size: Round joins count their arc points on every instance's transformed polyline, so the
    sizes are the instance's (on this drawing -- thin strokes over gently turning polylines -- every join's arc happens to be one segment,
    so the instances come out equal; instances of different sizes: tests/test_gpu_tmpl.py); template mode with per-step sizes
    (k_tmpl_round_sizes + k_tmpl_emit_round). Digests of positions / colours
    / indices and the sizes of all 10 000 instances against the reference's, and the caller's mesh table against the streams."""
    import torch
    K = 10000
    ps, ops = wl.tiger_paths()
    P = len(ops)
    d = wl.tiger_draws(ops, K, join=1)
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    assert ctx.failure_info()["segment_items"] == 5  # template mode
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    bufs.pos.fill_(float("nan"))
    ctx.set_profiling(True)
    rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert [n for n, _ in ctx.stage_times()] == ["tmpl_round_sizes", "tmpl_emit"]
    ctx.set_profiling(False)
    assert int(bufs.dev_status.item()) == 0
    ref = _reference_rows("tigerround", K)  # [K, 12 + 2]
    mt = bufs.meshes[:nm * 32].view(torch.int64).view(-1, 4)
    draw = (mt[:, 3] & 0xFFFFFFFF).contiguous()
    m0 = torch.searchsorted(draw, torch.arange(K + 1, dtype=torch.int64, device=draw.device) * P)  # first mesh of every instance
    fvm = torch.cat([mt[:, 0], torch.tensor([nv], dtype=torch.int64, device=draw.device)])
    fim = torch.cat([mt[:, 1], torch.tensor([ni], dtype=torch.int64, device=draw.device)])
    # the table is consistent with itself: every mesh begins where the one in front of it ends
    assert torch.equal(fvm[1:], fvm[:-1] + (mt[:, 2] & 0xFFFFFFFF)) and torch.equal(fim[1:], fim[:-1] + ((mt[:, 2] >> 32) & 0xFFFFFFFF))
    fv, fi = fvm[m0[:-1]], fim[m0[:-1]]
    cv, ci = fvm[m0[1:]] - fv, fim[m0[1:]] - fi
    assert np.array_equal(cv.cpu().numpy(), ref[:, 12]) and np.array_equal(ci.cpu().numpy(), ref[:, 13])
    got = np.concatenate([hu.digest_ragged_torch(bufs.pos[:nv].view(torch.int32), 2 * fv, 2 * cv), hu.digest_ragged_torch(bufs.color[:nv], fv, cv),
                          hu.digest_ragged_torch(bufs.idx[:ni], fi, ci, is_u16=True)], axis=1)
    bad = np.flatnonzero((got != ref[:, :12]).any(axis=1))
    assert bad.shape[0] == 0, ("instances that differ from the reference", bad[:10].tolist(), bad.shape[0])
    pset.close()
    ctx.close()


@pytest.mark.parametrize("which", ["tigerroundwide", "variedround", "variedround_ordinary"])
def test_every_instance_with_round_joins_of_different_sizes_matches_the_reference(rt, wl, which, monkeypatch):
    """Round joins where the sizes really differ, at full size. tigerroundwide: Tiger x10k, strokes six times as wide and
    every instance stretched by its own (1 + e, 1 - e) (avgScale stays 1: ONE template class) -- the joins' arcs have different point counts
    from instance to instance, so the per-step sizes pass, the scan over all meshes and the device-side capacity check of the Round-join
    template see 10 000 instances of (33 distinct) sizes. variedround: Tiger x10k at seven scales (18 tolerance classes) with Round joins
    -- first through the ordinary pipeline (variedround_ordinary: VGX_TMPL_ROUND=0, k_flatten_inst + k_round_sizes + k_stroke), then through
    the class-aware Round-join templates of round 6 (one template per class, per-step tables addressed per instance).
    Digests of positions / colours / indices and the sizes of every instance against the reference's; the mesh table against the streams."""
    import torch
    K = 10000
    ordinary = which.endswith("_ordinary")
    if ordinary:
        monkeypatch.setenv("VGX_TMPL_ROUND", "0")
        which = which[:-len("_ordinary")]
    ps, ops = wl.tiger_paths()
    P = len(ops)
    if which == "tigerroundwide":
        d = wl.tiger_draws([dict(op, stroke_width=op["stroke_width"] * 6.0) for op in ops], K, join=1, stretch=True)
    else:
        d = wl.tiger_varied_draws(ops, K, join=1)
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    mode = ctx.failure_info()["segment_items"]
    assert mode == (4 if ordinary else 5), mode  # instances sorted by tolerance class / template mode
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    bufs.pos.fill_(float("nan"))
    ctx.set_profiling(True)
    rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    stages = [n for n, _ in ctx.stage_times()]
    ctx.set_profiling(False)
    if not ordinary:
        assert stages == ["tmpl_round_sizes", "tmpl_emit"], stages
    assert int(bufs.dev_status.item()) == 0
    ref = _reference_rows(which, K)  # [K, 12 + 2]
    assert len(np.unique(ref[:, 12])) > 8, "the instances differ in size"
    mt = bufs.meshes[:nm * 32].view(torch.int64).view(-1, 4)
    draw = (mt[:, 3] & 0xFFFFFFFF).contiguous()
    m0 = torch.searchsorted(draw, torch.arange(K + 1, dtype=torch.int64, device=draw.device) * P)  # first mesh of every instance
    fvm = torch.cat([mt[:, 0], torch.tensor([nv], dtype=torch.int64, device=draw.device)])
    fim = torch.cat([mt[:, 1], torch.tensor([ni], dtype=torch.int64, device=draw.device)])
    assert torch.equal(fvm[1:], fvm[:-1] + (mt[:, 2] & 0xFFFFFFFF)) and torch.equal(fim[1:], fim[:-1] + ((mt[:, 2] >> 32) & 0xFFFFFFFF))
    fv, fi = fvm[m0[:-1]], fim[m0[:-1]]
    cv, ci = fvm[m0[1:]] - fv, fim[m0[1:]] - fi
    assert np.array_equal(cv.cpu().numpy(), ref[:, 12]) and np.array_equal(ci.cpu().numpy(), ref[:, 13])
    got = np.concatenate([hu.digest_ragged_torch(bufs.pos[:nv].view(torch.int32), 2 * fv, 2 * cv), hu.digest_ragged_torch(bufs.color[:nv], fv, cv),
                          hu.digest_ragged_torch(bufs.idx[:ni], fi, ci, is_u16=True)], axis=1)
    bad = np.flatnonzero((got != ref[:, :12]).any(axis=1))
    assert bad.shape[0] == 0, ("instances that differ from the reference", bad[:10].tolist(), bad.shape[0])
    pset.close()
    ctx.close()


def test_every_mesh_of_the_round_join_polylines_matches_the_reference(rt, wl):
    """BASELINE configs[3]: 10 000 polylines x 1 000 segments, Round joins + Round caps (data-dependent mesh sizes)."""
    import torch
    ps, d = wl.random_walk_polylines(10000, 1000, seed=5678)
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    assert nm == 10000
    bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    ref = _reference_rows("round", 10000)  # [10000, 12 + 2]
    mt = bufs.meshes[:nm * 32].view(torch.int64).view(-1, 4)  # first_vertex, first_index, (num_vertices | num_indices << 32), (draw | kind << 32)
    fv, fi = mt[:, 0], mt[:, 1]
    cv, ci = mt[:, 2] & 0xFFFFFFFF, (mt[:, 2] >> 32) & 0xFFFFFFFF
    assert np.array_equal(cv.cpu().numpy(), ref[:, 12]) and np.array_equal(ci.cpu().numpy(), ref[:, 13])
    got = np.concatenate([hu.digest_ragged_torch(bufs.pos[:nv].view(torch.int32), 2 * fv, 2 * cv), hu.digest_ragged_torch(bufs.color[:nv], fv, cv),
                          hu.digest_ragged_torch(bufs.idx[:ni], fi, ci, is_u16=True)], axis=1)
    bad = np.flatnonzero((got != ref[:, :12]).any(axis=1))
    assert bad.shape[0] == 0, ("meshes that differ from the reference", bad[:10].tolist(), bad.shape[0])
    pset.close()
    ctx.close()


def test_every_path_of_the_million_cubics_matches_the_reference(rt, wl):
    """BASELINE configs[1]: 1 M independent cubics, flatten only (vgx_flatten_count + vgx_flatten_emit with transformPath)."""
    import torch
    ps, d = wl.random_cubics(1000000, seed=1234, box=1000.0)
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    r = rt.flatten(ctx, pset, dd, d.shape[0], apply_transform=True, to_host=False)
    ref = _reference_rows("cubics", 1000000)  # [1M, 4 + 1]
    di = r.dinfo_dev[:d.shape[0] * 40].view(torch.int64).view(-1, 5)  # first_poly_vertex, first_subpath, first_mesh, (num_poly_vertices | num_subpaths << 32), ...
    fv, cv = di[:, 0], di[:, 3] & 0xFFFFFFFF
    assert np.array_equal(cv.cpu().numpy(), ref[:, 4])
    npv = r.sizes["num_poly_vertices"]
    got = hu.digest_ragged_torch(r.poly_dev[:npv].view(torch.int32), 2 * fv, 2 * cv)
    bad = np.flatnonzero((got != ref[:, :4]).any(axis=1))
    assert bad.shape[0] == 0, ("paths that differ from the reference", bad[:10].tolist(), bad.shape[0])
    # the per-draw records and the sub-path records of all 1 M paths (rt.flatten returned vgx_flatten's -- the one-walk kernel's --
    # buffers after comparing them with the two-phase entry's): one open sub-path per path, places = the running sums of the counts
    n = d.shape[0]
    assert r.sizes["num_subpaths"] == n
    ex = torch.cumsum(cv, 0) - cv
    assert torch.equal(fv, ex) and torch.equal(di[:, 1], torch.arange(n, dtype=torch.int64, device=di.device)) and bool((di[:, 2] == 0).all())
    assert bool(((di[:, 3] >> 32) == 1).all()) and bool((di[:, 4] == 0).all())  # num_subpaths 1; num_meshes 0, flags 0 (no draw took the serial path)
    sp = r.subs_dev[:n * 16].view(torch.int64).view(-1, 2)  # first_vertex, (num_vertices | flags << 32)
    assert torch.equal(sp[:, 0], ex) and torch.equal(sp[:, 1], cv)
    pset.close()
    ctx.close()


@pytest.mark.parametrize("box", [10.0, 100.0, 10000.0])
def test_cubics_box_sweep_matches_the_reference(rt, wl, box):
    """SURVEY 8(d) config 2, "also run boxes 10 / 100 / 10 000 to sweep the output size": 250 000 cubics with coordinates in
    [0, box) -- ~5, ~14 and ~145 segments per cubic (the deepest trees the walk sees) -- every path against the reference
    (bench.py reports the same sweep at 1 M paths under configs.cubics1m.box_sweep)."""
    import torch
    paths = 250000
    ps, d = wl.random_cubics(paths, seed=1234, box=box)
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    r = rt.flatten(ctx, pset, dd, d.shape[0], apply_transform=True, to_host=False)
    ref = _reference_rows("cubics@%g:%d" % (box, paths), paths)  # [paths, 4 + 1]
    di = r.dinfo_dev[:d.shape[0] * 40].view(torch.int64).view(-1, 5)
    fv, cv = di[:, 0], di[:, 3] & 0xFFFFFFFF
    assert np.array_equal(cv.cpu().numpy(), ref[:, 4])
    npv = r.sizes["num_poly_vertices"]
    got = hu.digest_ragged_torch(r.poly_dev[:npv].view(torch.int32), 2 * fv, 2 * cv)
    bad = np.flatnonzero((got != ref[:, :4]).any(axis=1))
    assert bad.shape[0] == 0, ("paths that differ from the reference", bad[:10].tolist(), bad.shape[0])
    pset.close()
    ctx.close()
