"""vgx_flatten (csrc/vgx_flat1.hip): the ordered ONE-WALK flatten, single asynchronous call -- pathReset + path commands +
pathGetVertices / pathGetSubPaths (+ transformPath) for every draw, reference src/path.cpp:44-78, 86-201, 684-726,
src/vg.cpp:4957-4975.

Every caller of `rt.flatten` already compares vgx_flatten's bytes with the two-phase entry (vg-renderer_amd/runtime.py, entry
"both"), i.e. the whole flatten parity / golden / full-size suite pins it. This file adds what is specific to the new kernel: its
own comparison with the oracle, segments of several chunks (draws longer than 64 commands), the leaf list overflowing, deep and
degenerate cubics (the second run), serial paths between ordinary ones, empty paths / batches, the look-back across few and many
waves, and the capacity verdict."""
import importlib
import os

import numpy as np
import pytest

from util import assert_flat_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    return importlib.import_module("vg-renderer_amd.runtime")


def _ctx_with(rt, **env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return rt.Context(0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _one_walk(rt, ctx, ps, d, xform, **kw):
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    r = rt.flatten(ctx, pset, dd, d.shape[0], apply_transform=xform, entry="both", **kw)
    pset.close()
    return r


@pytest.mark.parametrize("box", [10.0, 1000.0, 10000.0])
@pytest.mark.parametrize("xform", [False, True])
def test_one_walk_random_cubics_vs_oracle(rt, gpu_ctx, wl, oracle, box, xform):
    ps, d = wl.random_cubics(6000, seed=99, box=box)
    if xform:
        d["mtx"][:, 0] = 0.75; d["mtx"][:, 1] = 0.25; d["mtx"][:, 2] = -0.5; d["mtx"][:, 3] = 1.25
        d["mtx"][:, 4] = np.arange(d.shape[0], dtype=np.float32) % 97; d["mtx"][:, 5] = 3.5
    got = _one_walk(rt, gpu_ctx, ps, d, xform)
    ref = oracle.flatten(ps, d, apply_transform=xform)
    assert_flat_equal(got, ref, "one-walk cubics box=%g xform=%s" % (box, xform))


@pytest.mark.parametrize("seed", [3, 4, 5, 6])
def test_one_walk_fuzz_all_commands_vs_oracle(rt, gpu_ctx, wl, oracle, seed):
    """Every path command, serial paths (arcs, closed shapes) between lane-parallel ones, fills and strokes enabled at random
    (the per-draw mesh counts / first_mesh come from the same look-back)."""
    ps = wl.fuzz_paths(300 + seed, npaths=160)
    d = wl.fuzz_draws(ps, 300 + seed)
    for xform in (False, True):
        got = _one_walk(rt, gpu_ctx, ps, d, xform)
        ref = oracle.flatten(ps, d, apply_transform=xform)
        assert_flat_equal(got, ref, "one-walk fuzz seed=%d xform=%s" % (seed, xform))


def _long_paths(rs, npaths, ncmd_lo, ncmd_hi, box=400.0):
    pm = importlib.import_module("vg-renderer_amd.pathset")
    b = pm.PathSetBuilder()
    for _ in range(npaths):
        b.begin_path()
        left = int(rs.randint(ncmd_lo, ncmd_hi))
        while left > 0:
            m = min(left, int(rs.randint(3, 40)))
            p = rs.uniform(0, box, size=2)
            b.move_to(*p)
            for _ in range(m):
                k = rs.randint(0, 4)
                q = rs.uniform(0, box, size=6)
                if k == 0:
                    b.line_to(q[0], q[1])
                elif k == 1:
                    b.quadratic_to(q[0], q[1], q[2], q[3])
                else:
                    b.cubic_to(*q)
            if rs.uniform() < 0.6:
                b.close()
            left -= m + 2
        b.end_path()
    return b.arrays()


@pytest.mark.parametrize("seed,lo,hi", [(1, 60, 70), (2, 65, 200), (3, 300, 900), (4, 1, 140)])
def test_one_walk_draws_longer_than_a_chunk(rt, gpu_ctx, wl, oracle, seed, lo, hi):
    """Segments of several 64-command chunks: counted first, published, walked again chunk by chunk; sub-paths and pathClose pops
    across chunk borders."""
    pm = importlib.import_module("vg-renderer_amd.pathset")
    rs = np.random.RandomState(seed)
    ps = _long_paths(rs, 40, lo, hi)
    d = pm.make_draws(300)
    d["path"] = rs.randint(0, ps.npaths, size=300)
    d["fill_flags"] = rs.choice([0, 1, 3], size=300)
    d["stroke_flags"] = rs.choice([0, 3], size=300)
    d["stroke_width"] = 2.0
    d["scale"] = rs.choice([0.5, 1.0, 2.0], size=300).astype(np.float32)
    d["mtx"][:, 4] = rs.uniform(-50, 50, size=300).astype(np.float32)
    for xform in (False, True):
        got = _one_walk(rt, gpu_ctx, ps, d, xform)
        ref = oracle.flatten(ps, d, apply_transform=xform)
        assert_flat_equal(got, ref, "long draws seed=%d xform=%s" % (seed, xform))


def test_one_walk_leaf_list_overflow_and_small_grids(rt, wl, oracle):
    """VGX_F1_CAP=1024: a chunk of 32 cubics x ~45 leaves does not fit the list -> its cubics are walked again, straight to memory.
    VGX_F1_WAVES=3 / 7: the look-back reaches far back (few waves, many tickets each) and must not depend on the grid."""
    ps, d = wl.random_cubics(20000, seed=7, box=1000.0)
    ref = oracle.flatten(ps, d, apply_transform=True)
    for env in ({"VGX_F1_CAP": 1024}, {"VGX_F1_WAVES": 3}, {"VGX_F1_WAVES": 7, "VGX_F1_CAP": 3072}, {"VGX_F1_WAVES": 4096}):
        ctx = _ctx_with(rt, **env)
        got = _one_walk(rt, ctx, ps, d, True)
        assert_flat_equal(got, ref, "one-walk %r" % (env,))
        ctx.close()


def test_one_walk_deep_and_degenerate_cubics(rt, gpu_ctx, wl, oracle):
    """Tolerances small enough for ten levels of subdivision (the LDS levels overflow -> full-depth redo; the reference's silent
    drop at depth 10, path.cpp:168-179) and cubics small enough for pathAddVertex's epsilon test (path.cpp:767-777): such draws
    are listed, counted by the exact serial builder and the kernel runs a second time."""
    pm = importlib.import_module("vg-renderer_amd.pathset")
    ps, d = wl.random_cubics(3000, seed=11, box=3000.0)
    d["tess_tol"][::7] = 1e-5     # very deep
    d["scale"][::11] = 16.0       # tol / scale^2
    ps2, d2 = wl.random_cubics(2000, seed=12, box=0.02)  # tiny: consecutive vertices closer than sqrt(1e-5)
    both = pm.concat([ps, ps2])
    d2["path"] += ps.npaths
    dd = np.concatenate([d, d2])
    rs = np.random.RandomState(5)
    dd = dd[rs.permutation(dd.shape[0])]
    got = _one_walk(rt, gpu_ctx, both, dd, True)
    ref = oracle.flatten(both, dd, apply_transform=True)
    assert got.sizes["num_serial_draws"] > 0
    assert_flat_equal(got, ref, "deep / degenerate cubics")


def test_one_walk_empty_paths_and_batches(rt, gpu_ctx, wl, oracle):
    pm = importlib.import_module("vg-renderer_amd.pathset")
    b = pm.PathSetBuilder()
    b.begin_path(); b.end_path()                                   # empty path
    b.begin_path(); b.move_to(1, 2); b.end_path()                  # one vertex
    b.begin_path(); b.move_to(0, 0); b.cubic_to(10, 30, 40, 30, 50, 0); b.close(); b.end_path()
    b.begin_path(); b.end_path()
    ps = b.arrays()
    d = pm.make_draws(200)
    d["path"] = np.arange(200) % 4
    d["fill_flags"] = 3
    got = _one_walk(rt, gpu_ctx, ps, d, False)
    ref = oracle.flatten(ps, d, apply_transform=False)
    assert_flat_equal(got, ref, "empty paths")
    d0 = pm.make_draws(8)
    d0["path"] = 0  # a batch without a single command
    got = _one_walk(rt, gpu_ctx, ps, d0, False)
    assert got.sizes["num_poly_vertices"] == 0 and got.sizes["num_subpaths"] == 0


def test_one_walk_capacity_verdict(rt, gpu_ctx, wl):
    import torch
    ps, d = wl.random_cubics(5000, seed=21, box=1000.0)
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    full = rt.flatten(gpu_ctx, pset, dd, d.shape[0], entry="two_phase")
    npv, nsp = full.sizes["num_poly_vertices"], full.sizes["num_subpaths"]
    for cap_poly, cap_subs in ((npv - 1, nsp), (npv, nsp - 1), (npv // 3, nsp)):
        bufs = rt.FlatBuffers(dd.device, cap_poly, cap_subs, d.shape[0])
        guard = torch.full((16, 2), 7.0, dtype=torch.float32, device=dd.device)
        rt.flatten_async(gpu_ctx, pset, dd, d.shape[0], bufs)
        torch.cuda.synchronize()
        assert int(bufs.dev_status.item()) == 4  # VGX_E_NOSPACE
        z = bufs.dev_sizes.cpu().numpy()
        assert int(z[0]) == npv and int(z[1]) == nsp  # the totals say what is needed
        assert bool((guard == 7.0).all())
    # and exact capacities pass
    bufs = rt.FlatBuffers(dd.device, npv, nsp, d.shape[0])
    rt.flatten_async(gpu_ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    assert torch.equal(bufs.poly[:npv].view(torch.int32), full.poly_dev[:npv].view(torch.int32))
    pset.close()


def test_one_walk_tiger_instances(rt, gpu_ctx, wl, oracle):
    """The tiger-like drawing (25-command draws, 1-3 closed sub-paths): most segments are one chunk, some two."""
    ps, d = wl.tiger(40)
    got = _one_walk(rt, gpu_ctx, ps, d, True)
    ref = oracle.flatten(ps, d, apply_transform=True)
    assert_flat_equal(got, ref, "tiger x40")


def test_one_walk_long_polylines(rt, gpu_ctx, wl, oracle):
    ps, d = wl.random_walk_polylines(n=60, nseg=1000, seed=5)
    got = _one_walk(rt, gpu_ctx, ps, d, True)
    ref = oracle.flatten(ps, d, apply_transform=True)
    assert_flat_equal(got, ref, "long polylines")


@pytest.mark.parametrize("box,n", [(10.0, 40000), (1000.0, 20000), (10000.0, 6000)])
def test_one_walk_repeated_calls_adapt_and_stay_exact(rt, wl, box, n):
    """vgx_flatten chooses its kernel instance / segment size (and, for very short curves, the two-walk kernels) from what the LAST
    call on the same batch produced: the first, second and third call on one path set take different routes and must write the
    same bytes."""
    import torch
    ps, d = wl.random_cubics(n, seed=31, box=box)
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    ref = rt.flatten(ctx, pset, dd, n, apply_transform=True, entry="two_phase", to_host=False)
    npv, nsp = ref.sizes["num_poly_vertices"], ref.sizes["num_subpaths"]
    for call in range(4):
        bufs = rt.FlatBuffers(dd.device, npv, nsp, n)
        rt.flatten_async(ctx, pset, dd, n, bufs, apply_transform=True)
        torch.cuda.synchronize()
        assert int(bufs.dev_status.item()) == 0, call
        z = bufs.dev_sizes.cpu().numpy()
        assert int(z[0]) == npv and int(z[1]) == nsp, call
        assert torch.equal(bufs.poly[:npv].view(torch.int32), ref.poly_dev[:npv].view(torch.int32)), call
        assert torch.equal(bufs.subs[:nsp * 16], ref.subs_dev[:nsp * 16]), call
        assert torch.equal(bufs.dinfo[:n * 40], ref.dinfo_dev[:n * 40]), call
    # a capacity that is too small is still reported on the adapted route
    small = rt.FlatBuffers(dd.device, npv - 1, nsp, n)
    rt.flatten_async(ctx, pset, dd, n, small, apply_transform=True)
    torch.cuda.synchronize()
    assert int(small.dev_status.item()) == 4
    pset.close()
    ctx.close()


def test_flatten_leaves_the_tessellate_guard_alone(rt, wl):
    """vgx_flatten used to overwrite the context's cmd_instances cap for good, so a later single-call
    vgx_tessellate on a batch that GREW in commands wrote its per-command scratch out of bounds instead of ending with
    VGX_E_NOSPACE. The cap is the last count's again after any vgx_flatten call (both of its routes)."""
    import torch
    ctx = rt.Context(0)
    ps, d = wl.tiger(3)
    pset = rt.PathSet(ctx, ps)
    small = np.ascontiguousarray(d[:40])
    dd_small, dd_all = rt.upload_draws(small), rt.upload_draws(d)
    a = rt.tessellate(ctx, pset, dd_small, small.shape[0])          # sizes the scratch for 40 draws
    ctx2 = rt.Context(0)
    pset2 = rt.PathSet(ctx2, ps)
    whole = rt.tessellate(ctx2, pset2, dd_all, d.shape[0])
    pset2.close()
    ctx2.close()
    big = rt.MeshBuffers(dd_all.device, whole.sizes["num_vertices"], whole.sizes["num_indices"], whole.sizes["num_meshes"])
    for rep in range(3):                                              # first call: one-walk route; later calls may take the two-walk shortcut
        fb = rt.FlatBuffers(dd_all.device, 4_000_000, 200_000, d.shape[0])
        rt.flatten_async(ctx, pset, dd_all, d.shape[0], fb, apply_transform=True)
        torch.cuda.synchronize()
        assert int(fb.dev_status.item()) == 0
    # the 720-draw batch through the scratch of the 40-draw count: reported, not overrun (capDraws is the first line of defence on the host)
    try:
        rt.tessellate_async(ctx, pset, dd_all, d.shape[0], big)
        torch.cuda.synchronize()
        assert int(big.dev_status.item()) == 4  # VGX_E_NOSPACE
    except rt.VgxError as e:
        assert e.status == 4
    # and the small batch still tessellates, bit for bit
    bufs = rt.MeshBuffers(dd_small.device, a.sizes["num_vertices"], a.sizes["num_indices"], a.sizes["num_meshes"])
    rt.tessellate_async(ctx, pset, dd_small, small.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    assert np.array_equal(bufs.pos[:a.sizes["num_vertices"]].cpu().numpy().view(np.uint32), a.pos.view(np.uint32))
    ctx.close()


def test_two_walk_shortcut_with_changed_paths(rt, wl):
    """the two-walk shortcut of vgx_flatten sized its per-command words from the LAST call's command total; a draw list
    of the same length that picks longer paths then ended with a spurious VGX_E_NOSPACE. Sized for any draw list of that length now."""
    import torch
    ctx = rt.Context(0)
    b = importlib.import_module("vg-renderer_amd").PathSetBuilder()
    for k in range(64):  # short curves (a few segments per cubic: the shortcut's territory); path k has 1 + k % 3 cubics
        b.begin_path()
        b.move_to(0.0, 0.0)
        for c in range(1 + k % 3):
            b.cubic_to(1.0 + c, 0.5, 2.0 + c, 0.5, 3.0 + c, 0.0)
        b.end_path()
    ps = b.arrays()
    n = 20000
    d = importlib.import_module("vg-renderer_amd").make_draws(n)
    d["path"] = (np.arange(n) * 3) % 64  # paths with ONE cubic only
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    fb = rt.FlatBuffers(dd.device, 2_000_000, 100_000, n)
    for _ in range(3):
        rt.flatten_async(ctx, pset, dd, n, fb, apply_transform=False)
        torch.cuda.synchronize()
        assert int(fb.dev_status.item()) == 0
    d2 = d.copy()
    d2["path"] = (np.arange(n) * 3 + 2) % 64  # same number of draws, three cubics each
    dd2 = rt.upload_draws(d2)
    rt.flatten_async(ctx, pset, dd2, n, fb, apply_transform=False)
    torch.cuda.synchronize()
    assert int(fb.dev_status.item()) == 0, "a draw list of the same length with longer paths"
    ref = rt.flatten(ctx, pset, dd2, n, apply_transform=False, entry="two_phase")
    z = fb.dev_sizes.cpu().numpy()
    assert int(z[0]) == ref.sizes["num_poly_vertices"]
    assert torch.equal(fb.poly[:int(z[0])].view(torch.int32), ref.poly_dev[:int(z[0])].view(torch.int32))
    pset.close()
    ctx.close()
