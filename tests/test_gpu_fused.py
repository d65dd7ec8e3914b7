"""GPU parity of the FUSED single-pass kernel (vgx_fused.hip) = what vgx_tessellate runs in steady state: flatten ->
transform -> stroker in one kernel with the polyline in LDS. Same bar as everywhere: mesh tables, indices, colours and
positions bit-exact against the CPU oracle (the reference's own sources when oracle/_ref is present)."""
import importlib

import numpy as np
import pytest

from util import assert_mesh_equal, describe_mesh_diff, run_async

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    return importlib.import_module("vg-renderer_amd.runtime")


@pytest.fixture(scope="module")
def gpu_ctx(rt):
    """The single-pass kernel is opt-in (VGX_FUSED=1, read at vgx_create): these tests run on their own context."""
    import os
    old = os.environ.get("VGX_FUSED")
    os.environ["VGX_FUSED"] = "1"
    ctx = rt.Context(0)
    if old is None:
        del os.environ["VGX_FUSED"]
    else:
        os.environ["VGX_FUSED"] = old
    yield ctx
    ctx.close()


def _check(got, ref, what, expect_fused=True):
    assert got.status == 0, (what, "status", got.status, got.failure)
    if expect_fused and got.stages is not None:
        assert "fused" in got.stages, (what, "the single-pass kernel did not run", got.stages)
    try:
        assert_mesh_equal(got, ref, what)
    except AssertionError as e:
        raise AssertionError("%s\n%r" % (e, describe_mesh_diff(got, ref)))
    for k in ("num_meshes", "num_vertices", "num_indices", "num_poly_vertices", "num_subpaths"):
        assert got.dev_sizes[k] == ref.sizes[k], (what, "device totals", k, got.dev_sizes, ref.sizes)


def test_fused_config0(rt, gpu_ctx, wl, oracle):
    ps, d = wl.single_cubic()
    _check(run_async(rt, gpu_ctx, ps, d, profile=True), oracle.tessellate(ps, d), "config0")


@pytest.mark.parametrize("k", [1, 3, 24])
def test_fused_tiger(rt, gpu_ctx, wl, oracle, k):
    ps, d = wl.tiger(k)
    _check(run_async(rt, gpu_ctx, ps, d, profile=True), oracle.tessellate(ps, d), "tiger x%d" % k)


@pytest.mark.parametrize("seed", list(range(100, 124)))
def test_fused_fuzz_all_commands_all_strokers(rt, gpu_ctx, wl, oracle, seed):
    """Every path command (arcs / shapes = exact one-lane-per-draw rebuild of the segment), degenerate input (epsilon
    de-dup, zero-length steps), every cap / join / AA / thin combination."""
    ps = wl.fuzz_paths(seed, npaths=96)
    d = wl.fuzz_draws(ps, seed)
    _check(run_async(rt, gpu_ctx, ps, d, profile=True), oracle.tessellate(ps, d), "fuzz seed=%d" % seed)


@pytest.mark.parametrize("seed", list(range(300, 312)))
def test_fused_fuzz_curves_only(rt, gpu_ctx, wl, oracle, seed):
    """No shapes / arcs: the lane-parallel walk is the path taken (degenerate draws still go through the rebuild)."""
    ps = wl.fuzz_paths(seed, npaths=200, with_shapes=False, degenerate=(seed & 1) == 0)
    d = wl.fuzz_draws(ps, seed, ndraws=1000)
    _check(run_async(rt, gpu_ctx, ps, d, profile=True), oracle.tessellate(ps, d), "curves seed=%d" % seed)


@pytest.mark.parametrize("cap,join", [(0, 0), (1, 1), (2, 2), (1, 0), (0, 1)])
def test_fused_long_polylines_in_window(rt, gpu_ctx, wl, oracle, cap, join):
    """1001-vertex polylines (one draw = one segment, the polyline just fits the 1024-vertex LDS window); Round joins
    are sized from the window."""
    ps, d = wl.random_walk_polylines(n=40, nseg=1000, seed=5678, cap=cap, join=join)
    _check(run_async(rt, gpu_ctx, ps, d, profile=True), oracle.tessellate(ps, d), "polylines cap=%d join=%d" % (cap, join))


@pytest.mark.parametrize("cap,join,aa", [(1, 1, True), (0, 0, True), (2, 2, False)])
def test_fused_polylines_through_the_heap(rt, gpu_ctx, wl, oracle, cap, join, aa):
    """1501-vertex polylines do not fit the LDS window: flattened a second time into a heap block, elements read HBM."""
    ps, d = wl.random_walk_polylines(n=30, nseg=1500, seed=91, cap=cap, join=join)
    if not aa:
        d["stroke_flags"] &= ~np.uint32(rt.capi.STROKE_AA)
    _check(run_async(rt, gpu_ctx, ps, d, profile=True), oracle.tessellate(ps, d), "heap polylines cap=%d join=%d aa=%s" % (cap, join, aa))


def test_fused_big_cubics(rt, gpu_ctx, wl, oracle):
    """Random cubics in a 1000-unit box: tens of leaves per cubic (leaf slots overflow to the wave's global area, deep
    cubics are redone with the full-depth stack), filled + stroked."""
    ps, d = wl.random_cubics(3000, seed=77, box=1000.0)
    wl.set_fill(d, slice(None), 0xFF336699, aa=True)
    wl.set_stroke(d, slice(None), 0xFF2080FF, 2.0, 0, 0, aa=True)
    _check(run_async(rt, gpu_ctx, ps, d, profile=True), oracle.tessellate(ps, d), "big cubics")


def test_fused_shapes_batch(rt, gpu_ctx, wl, oracle):
    """Rounded rects / circles / ellipses only: every segment is rebuilt by the exact sequential builder."""
    pb = importlib.import_module("vg-renderer_amd.pathset")
    rs = np.random.RandomState(5)
    b = pb.PathSetBuilder()
    n = 600
    for i in range(n):
        b.begin_path()
        k = i % 3
        if k == 0:
            b.rounded_rect(rs.uniform(0, 500), rs.uniform(0, 500), rs.uniform(5, 80), rs.uniform(5, 80), rs.uniform(0, 20))
        elif k == 1:
            b.circle(rs.uniform(0, 500), rs.uniform(0, 500), rs.uniform(1, 60))
        else:
            b.ellipse(rs.uniform(0, 500), rs.uniform(0, 500), rs.uniform(1, 60), rs.uniform(1, 60))
        b.end_path()
    ps = b.arrays()
    d = pb.make_draws(n)
    d["path"] = np.arange(n, dtype=np.uint32)
    wl.set_fill(d, slice(None), 0xFF808080, aa=True)
    wl.set_stroke(d, slice(None), 0xFF101010, 1.5, 1, 1, aa=True)
    _check(run_async(rt, gpu_ctx, ps, d, profile=True), oracle.tessellate(ps, d), "shapes")


@pytest.mark.parametrize("waves", ["1", "3", "64"])
def test_fused_few_waves(rt, wl, oracle, waves, monkeypatch):
    """The look-back with one wave (every predecessor already has its prefix), three, and sixty-four waves."""
    monkeypatch.setenv("VGX_FUSED_WAVES", waves)
    monkeypatch.setenv("VGX_FUSED", "1")
    ctx = rt.Context(0)
    ps, d = wl.tiger(24)
    _check(run_async(rt, ctx, ps, d, profile=True), oracle.tessellate(ps, d), "tiger x24, %s waves" % waves)
    ps = wl.fuzz_paths(411, npaths=128)
    d = np.concatenate([wl.fuzz_draws(ps, 411)] * 6)
    _check(run_async(rt, ctx, ps, d, profile=True), oracle.tessellate(ps, d), "fuzz x6, %s waves" % waves)
    ctx.close()


def test_fused_reports_small_output_buffers(rt, gpu_ctx, wl):
    ps, d = wl.tiger(4)
    got = run_async(rt, gpu_ctx, ps, d, shrink=0.5)
    assert got.status == rt.capi.VGX_E_NOSPACE


def test_fused_matches_multi_kernel_pipeline(rt, gpu_ctx, wl, monkeypatch):
    """Same batch through vgx_tessellate with the fused kernel and (own context, VGX_NO_FUSED) the multi-kernel pipeline."""
    ps, d = wl.tiger(200)
    a = run_async(rt, gpu_ctx, ps, d, profile=True)
    assert "fused" in a.stages
    monkeypatch.delenv("VGX_FUSED", raising=False)
    ctx2 = rt.Context(0)
    b = run_async(rt, ctx2, ps, d, profile=True)
    assert "fused" not in b.stages and "flatten_build" in b.stages
    ctx2.close()
    assert a.status == 0 and b.status == 0
    assert np.array_equal(a.idx, b.idx) and np.array_equal(a.color, b.color)
    assert np.array_equal(a.pos.view(np.uint32), b.pos.view(np.uint32))
    for k in a.meshes.dtype.names:
        assert np.array_equal(a.meshes[k], b.meshes[k]), k
