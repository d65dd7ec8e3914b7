"""GPU parity tests of the instanced flatten kernel (k_flatten_inst, csrc/vgx_inst.hip): batches whose draws repeat one
sequence of paths (draws[i].path == draws[i mod P].path, at least 32 repetitions, more than VGX_SMALL_DRAWS draws) are
flattened with one lane per instance. Every case goes through the C-ABI entry point bench.py times (vgx_tessellate)
and is compared with the reference oracle on the complete output (sizes, mesh table, positions at 0 ulp, colours,
indices). Options are read at vgx_create, so cases with knobs own their context."""
import numpy as np
import pytest

from util import assert_mesh_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import importlib
    return importlib.import_module("vg-renderer_amd.runtime")


def _async(rt, ctx, ps, d, d_steady=None):
    """vgx_tessellate_count on d, then vgx_tessellate on d_steady (default: d). Returns the result + device totals."""
    import torch
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    bufs.pos.fill_(float("nan"))
    if d_steady is not None:
        dd.copy_(torch.from_numpy(d_steady.view(np.uint8).reshape(-1).copy()))
    rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()

    class G:
        pass
    got = G()
    got.status = int(bufs.dev_status.item())
    got.dev_sizes = bufs.dev_sizes.cpu().numpy().view(np.uint64)
    got.sizes = sizes
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    got.pos = bufs.pos[:nv].cpu().numpy()
    got.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
    got.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
    got.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    pset.close()
    return got


NUM_SERIAL = 5  # index of num_serial_draws in vgx_sizes


def _instances(wl, ps, seed, ninst, vary=True):
    """`ninst` repetitions of one fuzz drawing; every instance gets its own transform and -- with `vary` -- its own
    scale / tolerance (different subdivision depth per lane), and a few instances drop the fill or the stroke."""
    rs = np.random.RandomState(seed + 1000)
    base = wl.fuzz_draws(ps, seed)
    P = base.shape[0]
    d = np.tile(base, ninst)
    for i in range(ninst):
        s = slice(i * P, (i + 1) * P)
        if vary:
            f = float(rs.choice([1.0, 1.0, 0.5, 2.0, 3.5]))
            ang = rs.uniform(0, 2 * np.pi)
            c, sn = np.float32(np.cos(ang)), np.float32(np.sin(ang))
            m = d["mtx"][s].copy()
            d["mtx"][s, 0] = (m[:, 0] * c - m[:, 1] * sn) * np.float32(f)
            d["mtx"][s, 1] = (m[:, 0] * sn + m[:, 1] * c) * np.float32(f)
            d["mtx"][s, 2] = (m[:, 2] * c - m[:, 3] * sn) * np.float32(f)
            d["mtx"][s, 3] = (m[:, 2] * sn + m[:, 3] * c) * np.float32(f)
            d["scale"][s] *= np.float32(f)
            if rs.uniform() < 0.3:
                d["tess_tol"][s] = np.float32(rs.choice([0.05, 0.25, 1.0]))
            if rs.uniform() < 0.15:
                d["fill_flags"][s] = 0
            if rs.uniform() < 0.15:
                d["stroke_flags"][s] &= ~np.uint32(1)
        d["mtx"][s, 4] += np.float32(37.0 * (i % 10))
        d["mtx"][s, 5] += np.float32(41.0 * (i // 10))
    return d


@pytest.mark.parametrize("seed,ninst,shapes", [(700, 64, False), (701, 70, False), (702, 33, True), (703, 130, True)])
def test_instanced_fuzz(rt, gpu_ctx, wl, oracle, seed, ninst, shapes):
    """Every command, degenerate steps inside the epsilon ball, closing onto the start point; full and partly filled
    last instance groups; per-instance scale / tolerance so that the lanes of a wave subdivide to different depths.
    Without shapes no draw may reach the exact serial kernel (the instanced lane IS the sequential algorithm)."""
    ps = wl.fuzz_paths(seed, npaths=72, with_shapes=shapes, with_polylines=True)
    d = _instances(wl, ps, seed, ninst)
    assert d.shape[0] > 2048
    ref = oracle.tessellate(ps, d)
    got = _async(rt, gpu_ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "instanced fuzz seed=%d x%d" % (seed, ninst))
    if not shapes:
        assert int(got.dev_sizes[NUM_SERIAL]) == 0


def test_instanced_is_what_runs_and_the_knob_turns_it_off(rt, wl, oracle, monkeypatch):
    """Degenerate draws (epsilon de-duplication hits) go to the serial kernel on the command-parallel path and stay in
    their lane on the instanced path: num_serial_draws tells which kernel built the batch."""
    ps = wl.fuzz_paths(710, npaths=72, with_shapes=False, with_polylines=True)
    d = _instances(wl, ps, 710, 48, vary=False)
    ref = oracle.tessellate(ps, d)
    monkeypatch.setenv("VGX_TMPL", "0")  # (round 5: the batch -- Round joins among its strokes -- would otherwise be a template batch)
    ctx = rt.Context(0)
    got = _async(rt, ctx, ps, d)
    assert got.status == 0 and int(got.dev_sizes[NUM_SERIAL]) == 0
    assert_mesh_equal(got, ref, "instanced (default)")
    ctx.close()
    monkeypatch.setenv("VGX_INST", "0")
    ctx = rt.Context(0)
    got = _async(rt, ctx, ps, d)
    assert got.status == 0 and int(got.dev_sizes[NUM_SERIAL]) > 0
    assert_mesh_equal(got, ref, "VGX_INST=0")
    ctx.close()


@pytest.mark.parametrize("block,waves", [("1", "3"), ("8", "5"), ("64", "4096")])
def test_instanced_small_lane_blocks(rt, wl, oracle, block, waves, monkeypatch):
    """Lane-private heap blocks of 1 / 8 / 64 vertices: every sub-path is moved (many times) while it grows, lanes of a
    wave allocate in different rounds, a few waves walk many instance groups each."""
    monkeypatch.setenv("VGX_INST_BLOCK", block)
    monkeypatch.setenv("VGX_INST_WAVES", waves)
    ctx = rt.Context(0)
    ps = wl.fuzz_paths(720, npaths=60, with_shapes=True)
    d = _instances(wl, ps, 720, 80)
    ref = oracle.tessellate(ps, d)
    got = _async(rt, ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "lane blocks of %s vertices, %s waves" % (block, waves))
    ps, d = wl.tiger(40)
    ref = oracle.tessellate(ps, d)
    got = _async(rt, ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "tiger x40, lane blocks of %s" % block)
    ctx.close()


def test_instanced_long_subpaths_and_deep_cubics(rt, gpu_ctx, wl, oracle):
    """Sub-paths far longer than a lane block (geometric growth, sized through inst_long_subpath_vertices) and cubics that
    nest deeper than the LDS levels of the hot walk (full-depth redo), 40 instances each."""
    ps, d1 = wl.random_walk_polylines(n=60, nseg=700, seed=17, cap=0, join=0, width=3.0)
    d = np.tile(d1, 40)
    d["mtx"][:, 4] = np.repeat(np.arange(40, dtype=np.float32) * 3.0, d1.shape[0])
    ref = oracle.tessellate(ps, d)
    got = _async(rt, gpu_ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "60 x 701-vertex polylines x 40 instances")
    ps, d1 = wl.random_cubics(80, seed=78, box=1000.0)
    wl.set_fill(d1, slice(None), 0xFF336699, aa=True)
    wl.set_stroke(d1, slice(None), 0xFF2080FF, 2.0, 0, 0, aa=True)
    d = np.tile(d1, 36)
    tol = np.repeat(np.float32([0.25, 0.02, 0.002, 0.25] * 9), d1.shape[0])
    d["tess_tol"] = tol
    ref = oracle.tessellate(ps, d)
    got = _async(rt, gpu_ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "80 big cubics x 36 instances, tolerances 0.25 .. 0.002")


def test_instanced_period_is_rechecked_on_every_call(rt, gpu_ctx, wl, oracle):
    """The count pass finds the period; vgx_tessellate alone must notice when the draw records it is handed no longer
    repeat with it (permuted draws -> the command-parallel kernel builds the batch) and use it again when they do."""
    ps = wl.fuzz_paths(730, npaths=72, with_shapes=True)
    d = _instances(wl, ps, 730, 40)
    rs = np.random.RandomState(3)
    dperm = d[rs.permutation(d.shape[0])]
    ref = oracle.tessellate(ps, dperm)
    assert ref.sizes["num_vertices"] == oracle.tessellate(ps, d).sizes["num_vertices"]
    got = _async(rt, gpu_ctx, ps, d, d_steady=dperm)
    assert got.status == 0
    assert_mesh_equal(got, ref, "periodic at the count, permuted at the call")
    # same instances in another order of whole instances: still periodic, other offsets everywhere
    P = d.shape[0] // 40
    order = rs.permutation(40)
    dinst = np.concatenate([d[i * P:(i + 1) * P] for i in order])
    ref = oracle.tessellate(ps, dinst)
    got = _async(rt, gpu_ctx, ps, d, d_steady=dinst)
    assert got.status == 0
    assert_mesh_equal(got, ref, "instances reordered")


@pytest.mark.parametrize("seed,shapes", [(740, False), (741, True)])
def test_grouped_mode_shuffled_and_culled_instances(rt, gpu_ctx, wl, oracle, seed, shapes):
    """Paths reused by many draws WITHOUT a repeating sequence -- instances in shuffled draw order, a tenth of the draws
    culled, a second drawing's draws mixed in: the count pass chooses the grouped mode (draws sorted by path on the device
    at every call) and the instanced kernel still builds the batch. Without shapes no draw reaches the serial kernel."""
    ps = wl.fuzz_paths(seed, npaths=56, with_shapes=shapes, with_polylines=True)
    d = _instances(wl, ps, seed, 60)
    rs = np.random.RandomState(seed)
    keep = rs.uniform(size=d.shape[0]) > 0.1
    d = d[keep][rs.permutation(int(keep.sum()))]
    assert d.shape[0] > 2048
    ref = oracle.tessellate(ps, d)
    got = _async(rt, gpu_ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "grouped mode seed=%d" % seed)
    if not shapes:
        assert int(got.dev_sizes[NUM_SERIAL]) == 0  # the instanced lanes built it (k_flatten_build would list degenerate draws)
    # steady state on ANOTHER arrangement of the same size: the sort is redone by every call
    d2 = d[rs.permutation(d.shape[0])]
    d2["mtx"][:, 4] += np.float32(3.0)
    ref2 = oracle.tessellate(ps, d2)
    got2 = _async(rt, gpu_ctx, ps, d, d_steady=d2)
    assert got2.status == 0
    assert_mesh_equal(got2, ref2, "grouped mode, rearranged at the call")


def test_grouped_mode_is_not_chosen_for_one_off_paths(rt, gpu_ctx, wl, oracle):
    """A batch whose paths are used once or twice stays on the command-parallel kernel (degenerate draws are listed as
    serial there)."""
    ps = wl.fuzz_paths(750, npaths=1500, with_shapes=False, with_polylines=True)
    d = wl.fuzz_draws(ps, 750, ndraws=3000)
    ref = oracle.tessellate(ps, d)
    got = _async(rt, gpu_ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "one-off paths")
    assert int(got.dev_sizes[NUM_SERIAL]) > 0


def test_grouped_mode_small_lane_blocks_uneven_groups(rt, wl, oracle, monkeypatch):
    """Very different use counts per path (1 ... 300 draws), 8-vertex lane blocks, 5 waves."""
    monkeypatch.setenv("VGX_INST_BLOCK", "8")
    monkeypatch.setenv("VGX_INST_WAVES", "5")
    ctx = rt.Context(0)
    ps = wl.fuzz_paths(760, npaths=40, with_shapes=True)
    base = wl.fuzz_draws(ps, 760)
    rs = np.random.RandomState(760)
    reps = np.minimum(300, np.maximum(1, (rs.pareto(0.7, size=base.shape[0]) * 20).astype(np.int64)))
    d = np.repeat(base, reps)
    d["mtx"][:, 4] += rs.uniform(-50, 50, size=d.shape[0]).astype(np.float32)
    d = d[rs.permutation(d.shape[0])]
    assert d.shape[0] > 2048 and d.shape[0] / 40 >= 32
    ref = oracle.tessellate(ps, d)
    got = _async(rt, ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "uneven groups, small lane blocks")
    ctx.close()


def test_grouped_mode_path_set_larger_than_the_lds_table(rt, gpu_ctx, wl, oracle):
    """More than 4096 paths in the set: the counting sort's LDS table is a one-probe hash (slot = path mod 4096, first comer
    owns the slot, the others take the global counters). Used paths collide on purpose (p and p + 4096)."""
    ps = wl.fuzz_paths(770, npaths=4400, with_shapes=False, with_polylines=True)
    base = wl.fuzz_draws(ps, 770)
    rs = np.random.RandomState(770)
    hot = np.concatenate([np.arange(3, 3 + 40), np.arange(4099, 4099 + 24)])  # 3 .. 26 collide with 4099 .. 4122
    d = np.concatenate([np.repeat(base[hot], 64), base[rs.randint(0, 4400, size=40)]])
    d["mtx"][:, 4] += rs.uniform(-50, 50, size=d.shape[0]).astype(np.float32)
    d = d[rs.permutation(d.shape[0])]
    assert d.shape[0] > 2048 and d.shape[0] // len(np.unique(d["path"])) >= 32
    ref = oracle.tessellate(ps, d)
    got = _async(rt, gpu_ctx, ps, d)
    assert got.status == 0
    assert_mesh_equal(got, ref, "grouped mode, hashed LDS table")
    assert int(got.dev_sizes[NUM_SERIAL]) == 0  # built by the instanced lanes


def _continuous_scales(wl, ps, seed, ninst):
    """Instances with a scale of their own each (0.5 .. 3.5, continuous) and a rotation."""
    rs = np.random.RandomState(seed + 2000)
    base = wl.fuzz_draws(ps, seed)
    P = base.shape[0]
    d = np.tile(base, ninst)
    f = np.repeat(rs.uniform(0.5, 3.5, size=ninst).astype(np.float32), P)
    ang = np.repeat(rs.uniform(0, 2 * np.pi, size=ninst), P)
    c, sn = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    m = d["mtx"].copy()
    d["mtx"][:, 0] = (m[:, 0] * c - m[:, 1] * sn) * f
    d["mtx"][:, 1] = (m[:, 0] * sn + m[:, 1] * c) * f
    d["mtx"][:, 2] = (m[:, 2] * c - m[:, 3] * sn) * f
    d["mtx"][:, 3] = (m[:, 2] * sn + m[:, 3] * c) * f
    d["scale"] *= f
    return d


@pytest.mark.parametrize("classes,mode", [(None, 4), ("1", 1), ("4", 4), ("65536", 4), ("perm0", 3), ("perm2", 4)])
def test_instances_of_different_scales_are_sorted_by_tolerance_class(rt, wl, oracle, classes, mode, monkeypatch):
    """A periodic batch whose instances differ in scale: lane = instance in draw order would leave the lock-step walk at nearly
    every cubic, so the count pass keeps the periodic mapping but sorts the INSTANCES by tolerance class (flatten mode 4);
    VGX_INST_PERM=0 sorts the draws by (path, tolerance class) instead (mode 3, what non-periodic batches get), VGX_INST_CLASSES=1
    switches both off; few / very many classes only change which instances share a wave. Same bits in all cases."""
    monkeypatch.setenv("VGX_TMPL_ROUND", "0")  # (round 5: with its Round joins the uniform batch below would be a template batch; this test is about k_flatten_inst)
    if classes in ("perm0", "perm2"):  # perm2: the several-kernel form of the instance sort (used beyond 2^18 instances)
        monkeypatch.setenv("VGX_INST_PERM", classes[-1])
    elif classes is not None:
        monkeypatch.setenv("VGX_INST_CLASSES", classes)
    ps = wl.fuzz_paths(780, npaths=48, with_shapes=False, with_polylines=True)
    d = _continuous_scales(wl, ps, 780, 70)
    assert d.shape[0] > 2048
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context()
    try:
        got = _async(rt, ctx, ps, d)
        assert got.status == 0
        assert ctx.failure_info()["segment_items"] == mode
        assert_mesh_equal(got, ref, "tolerance classes=%s" % classes)
        assert int(got.dev_sizes[NUM_SERIAL]) == 0
        # the same context, a batch of ONE scale per path position: back to the periodic mapping
        d1 = _instances(wl, ps, 780, 70, vary=False)
        got1 = _async(rt, ctx, ps, d1)
        assert ctx.failure_info()["segment_items"] == 1
        assert_mesh_equal(got1, oracle.tessellate(ps, d1), "uniform scale after a varied batch")
    finally:
        ctx.close()


def test_tolerance_classes_with_outliers(rt, gpu_ctx, wl, oracle):
    """Shuffled instances of many scales plus a few draws whose tolerance is ten orders of magnitude away from the rest
    (1e-4 and 1e6): the classes quantise the RANGE of the batch, so most draws share a handful of classes then -- still
    the same bits. (tess_tol -> 0 is not a case: the reference's subdivision does not terminate in reasonable time.)"""
    ps = wl.fuzz_paths(781, npaths=40, with_shapes=False, with_polylines=True)
    d = _continuous_scales(wl, ps, 781, 64)
    rs = np.random.RandomState(781)
    d = d[rs.permutation(d.shape[0])]
    d["tess_tol"][rs.randint(0, d.shape[0], size=12)] = np.float32(1e-4)
    d["tess_tol"][rs.randint(0, d.shape[0], size=12)] = np.float32(1e6)
    ref = oracle.tessellate(ps, d)
    got = _async(rt, gpu_ctx, ps, d)
    assert got.status == 0
    assert gpu_ctx.failure_info()["segment_items"] == 3
    assert_mesh_equal(got, ref, "tolerance classes, outliers")
