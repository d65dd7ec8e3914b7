"""GPU parity tests of the TEMPLATE mode (csrc/vgx_tmpl.hip): a drawing submitted for >= 32 instances that differ only in
transform and colours is flattened ONCE, in local space, by vgx_tessellate_count; vgx_tessellate then transforms the
template's vertices per instance in registers and runs the stroker's per-element arithmetic on them (reference order of
operations: pathXXX in local space, transformPath, strokerXXX -- src/vg.cpp:4957-4975). Every case is compared with the
reference oracle on the complete output (sizes, mesh table, positions at 0 ulp, colours, indices) and, where it says so,
byte for byte with the ordinary pipeline (VGX_TMPL=0: k_flatten_inst + k_fill + k_stroke)."""
import numpy as np
import pytest

from util import assert_mesh_equal, bytes_equal

pytestmark = pytest.mark.gpu

MODE_TEMPLATE = 5
VGX_E_NOSPACE = 4
VGX_E_STALE = 10


@pytest.fixture(scope="module")
def rt():
    import importlib
    return importlib.import_module("vg-renderer_amd.runtime")


class _G:
    pass


ROUND_STAGES = ["tmpl_round_sizes", "tmpl_emit"]  # a step of a template with Round joins


def _run(rt, ctx, ps, d, d_steady=None, nd_steady=None, shrink=None, two_phase=False):
    """vgx_tessellate_count on d, then vgx_tessellate on d_steady (default d; nd_steady draws of it)."""
    import torch
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    mode = ctx.failure_info()["segment_items"]
    nd = d.shape[0] if nd_steady is None else nd_steady
    if d_steady is not None:
        dd = rt.upload_draws(d_steady)
    frac = nd / d.shape[0]
    nv, ni, nm = (int(round(sizes[k] * frac)) for k in ("num_vertices", "num_indices", "num_meshes"))
    if shrink:
        bufs = rt.MeshBuffers(dd.device, int(nv * shrink), ni, nm)
    else:
        bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    bufs.pos.fill_(float("nan"))
    bufs.idx.fill_(-1)
    bufs.color.fill_(0x5A5A5A5A)
    ctx.set_profiling(True)
    if two_phase:
        rt.tessellate_emit(ctx, pset, dd, nd, bufs)
    else:
        rt.tessellate_async(ctx, pset, dd, nd, bufs)
    torch.cuda.synchronize()
    g = _G()
    g.stages = [n for n, _ in ctx.stage_times()]
    ctx.set_profiling(False)
    g.mode = mode
    g.status = 0 if two_phase else int(bufs.dev_status.item())
    g.sizes = {"num_vertices": nv, "num_indices": ni, "num_meshes": nm}
    g.dev_sizes = bufs.dev_sizes.cpu().numpy().view(np.uint64)
    if not shrink:
        g.pos = bufs.pos[:nv].cpu().numpy()
        g.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
        g.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
        g.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    pset.close()
    return g


def test_tiger_template_is_what_runs_and_equals_reference_and_ordinary_path(rt, wl, oracle, monkeypatch):
    """The BASELINE drawing x 40 instances: template mode is chosen, its stages are the only ones that run, the output
    equals the reference's and the ordinary pipeline's (VGX_TMPL=0) byte for byte."""
    ps, d = wl.tiger(40)
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == ["tmpl_emit"], (got.mode, got.stages)
    assert got.status == 0
    assert_mesh_equal(got, ref, "tiger x40 template")
    assert int(got.dev_sizes[3]) == ref.sizes["num_vertices"] and int(got.dev_sizes[4]) == ref.sizes["num_indices"]
    small = ctx.scratch_bytes()
    ctx.close()
    monkeypatch.setenv("VGX_TMPL", "0")
    ctx = rt.Context(0)
    old = _run(rt, ctx, ps, d)
    assert old.mode != MODE_TEMPLATE and "tmpl_emit" not in old.stages
    for k in ("pos", "color", "idx", "meshes"):
        assert bytes_equal(getattr(got, k), getattr(old, k)), k
    assert small < ctx.scratch_bytes()  # no polyline heap, no per-command / per-mesh scratch for the whole batch
    ctx.close()


@pytest.mark.parametrize("seed,ninst,tile", [(900, 40, None), (901, 33, "64"), (902, 64, "128"), (903, 57, "960"), (904, 36, "192")])
def test_template_fuzz(rt, wl, oracle, monkeypatch, seed, ninst, tile):
    """Closed-shape fuzz drawings (every path command, serial shapes included; fills AA / plain / SSE index order; hairline
    and regular closed Miter strokes; per-path scale / tolerance / fringe) under random affine instance transforms --
    rotations, shears, mirrored instances (orientation and inner sides flip per instance) -- and per-instance colours;
    tile sizes that do and do not divide the element count (meshes cut by tile borders fetch their neighbours from L2)."""
    if tile:
        monkeypatch.setenv("VGX_TMPL_TILE", tile)
    ps = wl.closed_fuzz_paths(seed, npaths=72)
    d = wl.template_draws(ps, seed, ninst)
    assert d.shape[0] > 2048
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE, got.mode
    assert got.status == 0
    assert_mesh_equal(got, ref, "template fuzz seed=%d x%d" % (seed, ninst))
    ctx.close()


def test_template_two_phase_entry_and_other_instance_counts(rt, wl, oracle):
    """vgx_tessellate_emit after the count uses the template too; a later vgx_tessellate may bring any whole number of
    instances of the counted drawing (tiles of a frame) -- verified on the device against the saved first period."""
    ps = wl.closed_fuzz_paths(910, npaths=72)
    d = wl.template_draws(ps, 910, 48)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d, two_phase=True)
    assert got.mode == MODE_TEMPLATE and got.stages == ["tmpl_emit"]
    assert_mesh_equal(got, oracle.tessellate(ps, d), "two-phase")
    P = ps.npaths
    tile = d[11 * P:30 * P].copy()  # 19 instances, not the first ones
    got = _run(rt, ctx, ps, d, d_steady=tile, nd_steady=tile.shape[0])
    assert got.status == 0
    assert_mesh_equal(got, oracle.tessellate(ps, tile), "19-instance tile")
    ctx.close()


def test_template_transform_and_colour_changes_are_free_everything_else_is_stale(rt, wl, oracle):
    """Between the count and the step the caller may change transforms, colours and state keys; a change of any field the
    flattener or the mesh sizes depend on -- in any instance -- ends the step with VGX_E_STALE instead of wrong output."""
    ps = wl.closed_fuzz_paths(920, npaths=72)
    d = wl.template_draws(ps, 920, 40)
    ctx = rt.Context(0)
    d2 = d.copy()
    rs = np.random.RandomState(5)
    d2["mtx"] = rs.uniform(-3, 3, size=d2["mtx"].shape).astype(np.float32)
    d2["fill_color"] = rs.randint(0, 1 << 32, size=d.shape[0], dtype=np.uint64).astype(np.uint32)
    d2["stroke_color"] = rs.randint(0, 1 << 32, size=d.shape[0], dtype=np.uint64).astype(np.uint32)
    d2["state_key"] = 7
    got = _run(rt, ctx, ps, d, d_steady=d2)
    assert got.mode == MODE_TEMPLATE and got.status == 0
    assert_mesh_equal(got, oracle.tessellate(ps, d2), "new transforms / colours")
    n = d.shape[0]
    for field, where, value in (("scale", n - 5, np.float32(1.25)), ("tess_tol", n // 2, np.float32(0.3)), ("fringe", 3 * ps.npaths + 1, np.float32(0.75)),
                                ("stroke_width", n - 1, np.float32(2.5)), ("fill_flags", n // 3, np.uint32(0)), ("stroke_flags", 100 + ps.npaths, np.uint32(0)),
                                ("path", n - 2, np.uint32(0))):
        d3 = d.copy()
        if d3[field][where] == value:
            value = value + 1
        d3[field][where] = value
        got = _run(rt, ctx, ps, d, d_steady=d3)
        assert got.status == VGX_E_STALE, (field, got.status)
    d4 = d.copy()
    d4["mtx"][n - 7, 2] = np.float32("nan")
    assert _run(rt, ctx, ps, d, d_steady=d4).status == 3  # VGX_E_NONFINITE
    ctx.close()


def test_template_capacity_is_checked_on_the_device(rt, wl):
    ps, d = wl.tiger(34)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d, shrink=0.9)
    assert got.mode == MODE_TEMPLATE and got.status == VGX_E_NOSPACE
    assert int(got.dev_sizes[3]) == got.sizes["num_vertices"]  # the need is reported (the buffers held 90 % of it)
    ctx.close()


def test_batches_that_are_not_templates_take_the_ordinary_path(rt, wl, oracle):
    """Round joins in SEVERAL classes (their sizes are per instance; the per-step tables exist for one class only), every instance
    different: the ordinary pipeline, same results."""
    ps = wl.closed_fuzz_paths(930, npaths=72)
    ctx = rt.Context(0)
    base = wl.template_draws(ps, 930, 40)
    cases = []
    d = base.copy()
    sel = (d["stroke_flags"] & 1) != 0
    d["stroke_flags"][sel] |= np.uint32(rt.capi.JOIN_ROUND << 6)
    d["scale"][-ps.npaths:] *= np.float32(1.5)  # the last instance at another scale: a second class
    cases.append(("round joins in two classes", d))
    d = wl.template_draws(ps, 930, 80)
    d["scale"][::ps.npaths] *= (np.float32(1.0) + np.arange(80, dtype=np.float32) / np.float32(128.0))
    cases.append(("every instance at a scale of its own (more flavours than classes)", d))
    for name, d in cases:
        got = _run(rt, ctx, ps, d)
        assert got.mode != MODE_TEMPLATE and "tmpl_emit" not in got.stages, name
        assert got.status == 0, name
        assert_mesh_equal(got, oracle.tessellate(ps, d), name)
    # one instance at another scale: a second class (since the classes exist), still the same results
    d = base.copy()
    d["scale"][-ps.npaths:] *= np.float32(1.5)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.status == 0
    assert_mesh_equal(got, oracle.tessellate(ps, d), "one instance at another scale")
    ctx.close()


@pytest.mark.parametrize("round_joins", [False, True])
def test_template_tiles_with_many_small_meshes(rt, wl, oracle, round_joins):
    """Rectangles and triangles only: a 1024-element tile touches more meshes than the LDS record table holds and takes the
    per-lane fallback of k_tmpl_emit; mixed with larger shapes so that both forms run in one launch. round_joins: the same through
    k_tmpl_emit_round (places from the per-step tables in the fallback as well)."""
    pm = __import__("importlib").import_module("vg-renderer_amd.pathset")
    rs = np.random.RandomState(77)
    b = pm.PathSetBuilder()
    for p in range(90):
        b.begin_path()
        for s in range(int(rs.randint(2, 7)) if p < 70 else 1):
            x, y = rs.uniform(-200, 200, size=2)
            if p >= 70:
                b.circle(x, y, float(rs.uniform(30, 90)))
            elif rs.uniform() < 0.5:
                b.rect(x, y, float(rs.uniform(2, 30)), float(rs.uniform(2, 30)))
            else:
                b.move_to(x, y)
                b.line_to(x + float(rs.uniform(5, 20)), y + float(rs.uniform(-3, 3)))
                b.line_to(x + float(rs.uniform(-3, 3)), y + float(rs.uniform(5, 20)))
                b.close()
        b.end_path()
    ps = b.arrays()
    d = wl.template_draws(ps, 940, 35)
    if round_joins:
        sel = (d["stroke_flags"] & 1) != 0
        d["stroke_flags"][sel] |= np.uint32(rt.capi.JOIN_ROUND << 6)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.status == 0
    assert got.stages == (ROUND_STAGES if round_joins else ["tmpl_emit"])
    assert_mesh_equal(got, oracle.tessellate(ps, d), "small meshes")
    ctx.close()


def _assembled(rt, ctx, ps, d, max_vb, split_state):
    """count + vgx_tessellate with draw-command assembly armed (armed BEFORE the count)."""
    import torch
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    cmds = torch.zeros(200000 * 48, dtype=torch.uint8, device=dd.device)
    ncmd = torch.zeros(1, dtype=torch.int64, device=dd.device)
    ctx.set_assembly(cmds, max_vb, ncmd, split_state=split_state)
    try:
        sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
        mode = ctx.failure_info()["segment_items"]
        nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
        bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
        bufs.idx.fill_(-1)
        ctx.set_profiling(True)
        rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
        torch.cuda.synchronize()
        stages = [n for n, _ in ctx.stage_times()]
        ctx.set_profiling(False)
    finally:
        ctx.set_assembly(None)
    g = _G()
    g.mode, g.stages, g.status = mode, stages, int(bufs.dev_status.item())
    g.ncmd = int(ncmd.item())
    g.dev_sizes = bufs.dev_sizes.cpu().numpy().view(np.uint64)
    g.cmds = cmds[:g.ncmd * 48].cpu().numpy().view(rt.capi.drawcmd_dtype)
    g.pos = bufs.pos[:nv].cpu().numpy()
    g.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
    g.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
    g.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    pset.close()
    return g


@pytest.mark.parametrize("seed,ninst,max_vb,split", [(950, 40, 65536, False), (951, 36, 2048, True), (952, 50, 700, True)])
def test_template_mode_with_draw_command_assembly(rt, wl, oracle, monkeypatch, seed, ninst, max_vb, split):
    """vgx_set_assembly armed: the template pass writes the batch's mesh table for the partition kernels and adds every mesh's base
    inside its draw command to the indices it emits. Draw commands, rebased indices and vertex streams equal the ordinary
    pipeline's (VGX_TMPL=0) byte for byte and the oracle's assembly of the reference's meshes; state keys differ per instance
    (they are not part of the template)."""
    ps = wl.closed_fuzz_paths(seed, npaths=72)
    d = wl.template_draws(ps, seed, ninst)
    rs = np.random.RandomState(seed)
    d["state_key"] = np.repeat(rs.randint(0, 3, size=(d.shape[0] + 6) // 7), 7)[:d.shape[0]].astype(np.uint32)  # runs of 7 draws per state
    ctx = rt.Context(0)
    got = _assembled(rt, ctx, ps, d, max_vb, split)
    ctx.close()
    assert got.mode == MODE_TEMPLATE and got.stages[-1] == "tmpl_emit" and "assemble" in got.stages, (got.mode, got.stages)
    assert got.status == 0 and int(got.dev_sizes[9]) == got.ncmd
    monkeypatch.setenv("VGX_TMPL", "0")
    ctx = rt.Context(0)
    old = _assembled(rt, ctx, ps, d, max_vb, split)
    ctx.close()
    assert old.mode != MODE_TEMPLATE and old.status == 0
    assert got.ncmd == old.ncmd and bytes_equal(got.cmds, old.cmds)
    for k in ("pos", "color", "idx", "meshes"):
        assert bytes_equal(getattr(got, k), getattr(old, k)), k
    ref = oracle.tessellate(ps, d)
    keys = d["state_key"][ref.meshes["draw"]] if split else None
    st, rcmds, ridx = oracle.assemble(ref.meshes, ref.idx, max_vb, mesh_keys=keys)
    assert st == 0 and rcmds.shape[0] == got.ncmd
    assert np.array_equal(got.idx, ridx)


# ---- several classes: every instance repeats ONE OF a few flavours of the period ---------------------------------------------
@pytest.mark.parametrize("seed,ninst,ncls,tile", [(970, 48, 2, None), (971, 40, 5, "128"), (972, 64, 16, "960"), (974, 140, 64, None), (973, 36, 3, "64")])
def test_template_classes_fuzz(rt, wl, oracle, monkeypatch, seed, ninst, ncls, tile):
    """The same fuzz drawings in several flavours (own scales / tolerances / fringes / fill kinds / stroke widths per class, so
    the classes have different polylines, mesh counts and sizes), instances of the flavours mixed at random: one template per
    class, every instance emitted from its class's tables at its own place. == the reference, == the ordinary pipeline."""
    if tile:
        monkeypatch.setenv("VGX_TMPL_TILE", tile)
    ps = wl.closed_fuzz_paths(seed, npaths=72)
    d, pick = wl.template_class_draws(ps, seed, ninst, ncls)
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == ["tmpl_emit"], (got.mode, got.stages)
    assert got.status == 0
    assert_mesh_equal(got, ref, "template classes seed=%d x%d / %d classes" % (seed, ninst, ncls))
    assert int(got.dev_sizes[3]) == ref.sizes["num_vertices"] and int(got.dev_sizes[4]) == ref.sizes["num_indices"]
    # two-phase entry on the same context
    got2 = _run(rt, ctx, ps, d, two_phase=True)
    assert_mesh_equal(got2, ref, "template classes, two-phase")
    ctx.close()
    monkeypatch.setenv("VGX_TMPL_CLASSES", "0")
    ctx = rt.Context(0)
    old = _run(rt, ctx, ps, d)
    assert old.mode != MODE_TEMPLATE
    for k in ("pos", "color", "idx", "meshes"):
        assert bytes_equal(getattr(got, k), getattr(old, k)), k
    ctx.close()


def test_template_classes_tiger_at_seven_scales(rt, wl, oracle):
    """The bench's `tiger10k_varied` drawing (Tiger instances at 7 scales under rotations: 7 subdivisions, 7 sets of stroke
    widths) x 64: seven classes."""
    ps, ops = wl.tiger_paths()
    d = wl.tiger_varied_draws(ops, 64)
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == ["tmpl_emit"], (got.mode, got.stages)
    assert got.status == 0
    assert_mesh_equal(got, ref, "tiger at 7 scales")
    ctx.close()


def test_template_classes_stale_and_limits(rt, wl, oracle, monkeypatch):
    """An instance that changes its class after the count, or a different batch size: VGX_E_STALE / the ordinary path. More
    flavours than VGX_TMPL_MAX_CLASSES (64): the ordinary pipeline from the start."""
    ps = wl.closed_fuzz_paths(975, npaths=72)
    d, pick = wl.template_class_draws(ps, 975, 40, 3)
    P = ps.npaths
    ctx = rt.Context(0)
    moved = d.copy()
    a, b = int(np.flatnonzero(pick == 0)[-1]), int(np.flatnonzero(pick == 1)[-1])
    tmp = moved[a * P:(a + 1) * P].copy()
    moved[a * P:(a + 1) * P] = moved[b * P:(b + 1) * P]  # the two instances swap flavours: sizes change under the table's feet
    moved[b * P:(b + 1) * P] = tmp
    got = _run(rt, ctx, ps, d, d_steady=moved)
    assert got.mode == MODE_TEMPLATE and got.status == VGX_E_STALE
    # transforms and colours stay free
    free = d.copy()
    free["mtx"][:, 4] += np.float32(3.0)
    free["fill_color"] ^= np.uint32(0x00FF00FF)
    got = _run(rt, ctx, ps, d, d_steady=free)
    assert got.status == 0
    assert_mesh_equal(got, oracle.tessellate(ps, free), "classes: transforms / colours changed")
    ctx.close()
    many, _ = wl.template_class_draws(ps, 976, 100, 65)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, many)
    assert got.mode != MODE_TEMPLATE and got.status == 0
    assert_mesh_equal(got, oracle.tessellate(ps, many), "65 flavours: ordinary pipeline")
    ctx.close()


@pytest.mark.parametrize("seed,ninst,ncls,max_vb,split", [(980, 40, 3, 65536, False), (981, 36, 4, 2048, True)])
def test_template_classes_with_draw_command_assembly(rt, wl, oracle, monkeypatch, seed, ninst, ncls, max_vb, split):
    ps = wl.closed_fuzz_paths(seed, npaths=72)
    d, _ = wl.template_class_draws(ps, seed, ninst, ncls)
    ctx = rt.Context(0)
    rs = np.random.RandomState(seed)
    d["state_key"] = np.repeat(rs.randint(0, 3, size=(d.shape[0] + 6) // 7), 7)[:d.shape[0]].astype(np.uint32)
    a = _assembled(rt, ctx, ps, d, max_vb, split)
    assert a.mode == MODE_TEMPLATE and a.status == 0 and a.stages[-1] == "tmpl_emit", (a.mode, a.status, a.stages)
    ctx.close()
    monkeypatch.setenv("VGX_TMPL", "0")
    ctx = rt.Context(0)
    b = _assembled(rt, ctx, ps, d, max_vb, split)
    assert b.mode != MODE_TEMPLATE and b.status == 0
    ctx.close()
    assert a.ncmd == b.ncmd
    for k in ("pos", "color", "idx", "meshes", "cmds"):
        assert bytes_equal(getattr(a, k), getattr(b, k)), k


# ---- general strokes: everything but Round joins -------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,ninst,tile,closed_only", [(990, 40, None, False), (991, 36, "128", False), (992, 48, "960", True), (993, 33, "64", False)])
def test_template_general_strokes(rt, wl, oracle, monkeypatch, seed, ninst, tile, closed_only):
    """Open sub-paths with Butt / Square / Round caps, Bevel joins, non-AA and hairline strokes next to the closed Miter AA ones and
    the fills: all of them have closed-form sizes, so the batch is a template batch; the general element code (elem_geometry /
    elem_emit, what k_stroke runs) works on the staged vertices with bases and previous-element rails in closed form. == the
    reference, == the ordinary pipeline byte for byte."""
    if tile:
        monkeypatch.setenv("VGX_TMPL_TILE", tile)
    ps = wl.closed_fuzz_paths(seed, npaths=72) if closed_only else wl.fuzz_paths(seed, npaths=72, with_shapes=True, degenerate=False)
    d = wl.template_general_draws(ps, seed, ninst)
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == ["tmpl_emit"], (got.mode, got.stages)
    assert got.status == 0
    assert_mesh_equal(got, ref, "template general strokes seed=%d" % seed)
    ctx.close()
    monkeypatch.setenv("VGX_TMPL", "0")
    ctx = rt.Context(0)
    old = _run(rt, ctx, ps, d)
    assert old.mode != MODE_TEMPLATE
    for k in ("pos", "color", "idx", "meshes"):
        assert bytes_equal(getattr(got, k), getattr(old, k)), k
    ctx.close()


# ---- Round joins: sizes that belong to the instance --------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,ninst,tile,closed_only", [(995, 40, None, False), (1995, 36, "128", False), (2995, 48, "960", True), (3995, 33, "64", False), (4995, 64, "2048", True),
                                                         (5995, 80, None, False), (5996, 130, "192", True)])  # (>= 64 instances: the sizes pass places the meshes per instance)
def test_template_round_joins(rt, wl, oracle, monkeypatch, seed, ninst, tile, closed_only):
    """Round joins count their arc points on the TRANSFORMED polyline (stroker.cpp:1146, 1592): mesh sizes -- and every output place
    behind such a mesh -- differ from instance to instance. Template mode counts them per step (k_tmpl_round_sizes: the emit kernel's
    phases on the same staged values, sums instead of stores; places by scans) and emits with those places (k_tmpl_emit_round).
    == the reference, == the ordinary pipeline (VGX_TMPL_ROUND=0) byte for byte, mesh table included."""
    if tile:
        monkeypatch.setenv("VGX_TMPL_TILE", tile)
    ps = wl.closed_fuzz_paths(seed, npaths=72) if closed_only else wl.fuzz_paths(seed, npaths=72, with_shapes=True, degenerate=False)
    d = wl.template_general_draws(ps, seed, ninst, round_joins=True)
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == ROUND_STAGES, (got.mode, got.stages)
    assert got.status == 0
    assert_mesh_equal(got, ref, "template round joins seed=%d" % seed)
    ctx.close()
    monkeypatch.setenv("VGX_TMPL_ROUND", "0")
    ctx = rt.Context(0)
    old = _run(rt, ctx, ps, d)
    assert old.mode != MODE_TEMPLATE
    for k in ("pos", "color", "idx", "meshes"):
        assert bytes_equal(getattr(got, k), getattr(old, k)), k
    ctx.close()


@pytest.mark.parametrize("seed,ninst,ncls,tile,closed_only", [(6101, 70, 2, None, False), (6102, 96, 5, "128", False), (6103, 130, 16, "960", True), (6104, 80, 3, "64", False),
                                                              (6105, 200, 7, None, True)])
def test_template_round_joins_in_several_classes(rt, wl, oracle, monkeypatch, seed, ninst, ncls, tile, closed_only):
    """Round joins in a template of SEVERAL classes (round 6; /root/reference/src/stroker.cpp:1580-1691, arc count from the transformed
    geometry): every instance repeats one of a few flavours of the period (other scales, widths, tolerances and stroke styles per flavour),
    the sizes of its Round-join meshes are its own. One template per class, the per-step tables addressed per instance (VgxTmplInst::m /
    ::rel), the sizes pass in its workgroup-per-instance shape. == the reference, == the ordinary pipeline byte for byte, and the steady
    state follows other transforms (other arcs, other sizes)."""
    if tile:
        monkeypatch.setenv("VGX_TMPL_TILE", tile)
    ps = wl.closed_fuzz_paths(seed, npaths=72) if closed_only else wl.fuzz_paths(seed, npaths=72, with_shapes=True, degenerate=False)
    d, pick = wl.template_class_round_draws(ps, seed, ninst, ncls, closed_aa_only=closed_only)
    assert len(set(pick.tolist())) == ncls
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == ROUND_STAGES, (got.mode, got.stages)
    assert got.status == 0
    assert_mesh_equal(got, ref, "round joins in %d classes, seed=%d" % (ncls, seed))
    # other transforms in the steady state: the sizes are counted again, per instance
    d2 = d.copy()
    rs = np.random.RandomState(seed)
    ang = rs.uniform(0, 2 * np.pi, size=ninst)
    sx, sy = rs.uniform(0.6, 1.7, size=ninst), rs.uniform(0.6, 1.7, size=ninst)
    P = ps.npaths
    m = d2["mtx"].reshape(ninst, P, 6)
    m[:, :, 0] = (np.cos(ang) * sx)[:, None]; m[:, :, 1] = (np.sin(ang) * sx)[:, None]
    m[:, :, 2] = (-np.sin(ang) * sy)[:, None]; m[:, :, 3] = (np.cos(ang) * sy)[:, None]
    ref2 = oracle.tessellate(ps, d2)
    need = (int(ref2.pos.shape[0]), int(ref2.idx.shape[0]))
    assert need != (int(ref.pos.shape[0]), int(ref.idx.shape[0]))
    import torch
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    rt.tessellate_count(ctx, pset, dd, d.shape[0])
    dd2 = rt.upload_draws(d2)
    bufs = rt.MeshBuffers(dd2.device, need[0] + 64, need[1] + 64, int(ref2.meshes.shape[0]))
    rt.tessellate_async(ctx, pset, dd2, d2.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    ds = bufs.dev_sizes.cpu().numpy().view(np.uint64)
    assert (int(ds[3]), int(ds[4])) == need
    g = _G()
    g.pos = bufs.pos[:need[0]].cpu().numpy(); g.color = bufs.color[:need[0]].cpu().numpy().view(np.uint32)
    g.idx = bufs.idx[:need[1]].cpu().numpy().view(np.uint16); g.meshes = bufs.meshes[:ref2.meshes.shape[0] * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    g.sizes = {"num_vertices": need[0], "num_indices": need[1], "num_meshes": int(ref2.meshes.shape[0])}
    assert_mesh_equal(g, ref2, "round joins in classes, other transforms")
    pset.close()
    ctx.close()
    monkeypatch.setenv("VGX_TMPL_ROUND", "0")
    ctx = rt.Context(0)
    old = _run(rt, ctx, ps, d)
    assert old.mode != MODE_TEMPLATE
    for k in ("pos", "color", "idx", "meshes"):
        assert bytes_equal(getattr(got, k), getattr(old, k)), k
    ctx.close()


@pytest.mark.parametrize("ninst", [40, 70])  # (70: the per-instance shape of the sizes pass -- its own capacity check)
def test_template_round_joins_sizes_follow_the_transforms(rt, wl, oracle, ninst):
    """The steady-state call of a Round-join template with OTHER transforms than the count saw: other arcs, other sizes -- counted on the
    device for this very batch (dev_sizes), checked against the caller's capacities there (VGX_E_NOSPACE with the need; nothing written
    past the buffers), and equal to the reference's when they fit."""
    ps = wl.closed_fuzz_paths(5995, npaths=72)
    d = wl.template_general_draws(ps, 5995, ninst, round_joins=True)
    d2 = d.copy()
    d2["mtx"][:, :4] *= np.float32(0.37)  # smaller on screen: the same step angle spans the same arc... but rounding differs; and
    d2["mtx"][:, 1] += np.float32(0.21)   # a shear changes the angles between the segments themselves
    ref2 = oracle.tessellate(ps, d2)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d, d_steady=d2)  # buffers sized for d
    need = (int(ref2.pos.shape[0]), int(ref2.idx.shape[0]))
    assert need != (got.sizes["num_vertices"], got.sizes["num_indices"]), "the test wants a batch whose sizes differ from the counted one's"
    assert (int(got.dev_sizes[3]), int(got.dev_sizes[4])) == need, (got.dev_sizes, need)  # vgx_sizes: num_vertices, num_indices
    if need[0] > got.sizes["num_vertices"] or need[1] > got.sizes["num_indices"]:
        assert got.status == rt.capi.VGX_E_NOSPACE
        assert np.isnan(got.pos).all(), "nothing is written when the batch does not fit"
    else:
        assert got.status == 0
        assert bytes_equal(got.pos[:need[0]], ref2.pos) and bytes_equal(got.idx[:need[1]], ref2.idx)
    ctx.close()
    # and with room for it: the reference's bytes
    import torch
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    rt.tessellate_count(ctx, pset, dd, d.shape[0])
    dd2 = rt.upload_draws(d2)
    bufs = rt.MeshBuffers(dd2.device, need[0] + 64, need[1] + 64, int(ref2.meshes.shape[0]))
    rt.tessellate_async(ctx, pset, dd2, d2.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    g = _G()
    g.pos = bufs.pos[:need[0]].cpu().numpy(); g.color = bufs.color[:need[0]].cpu().numpy().view(np.uint32)
    g.idx = bufs.idx[:need[1]].cpu().numpy().view(np.uint16); g.meshes = bufs.meshes[:ref2.meshes.shape[0] * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    g.sizes = {"num_vertices": need[0], "num_indices": need[1], "num_meshes": int(ref2.meshes.shape[0])}
    assert_mesh_equal(g, ref2, "round joins, other transforms")
    pset.close()
    ctx.close()


def test_template_round_joins_tiger_stretched_wide(rt, wl, oracle):
    """The Tiger with Round joins, strokes six times as wide (arcs of several points) and every instance stretched by its own
    (1 + e, 1 - e) (avgScale stays 1): 300 instances of different sizes, tiles of 2048 elements with meshes that span tiles."""
    ps, ops = wl.tiger_paths()
    ops = [dict(op, stroke_width=op["stroke_width"] * 6.0) for op in ops]
    d = wl.tiger_draws(ops, 300, join=1, stretch=True)
    ref = oracle.tessellate(ps, d)
    P = len(ops)
    m0 = np.searchsorted(ref.meshes["draw"], np.arange(301, dtype=np.int64) * P)
    fv = np.concatenate([ref.meshes["first_vertex"].astype(np.int64), [ref.pos.shape[0]]])[m0]
    assert len(np.unique(np.diff(fv))) > 8, "the instances differ in size"
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == ROUND_STAGES and got.status == 0
    assert_mesh_equal(got, ref, "tiger, round joins, stretched")
    ctx.close()


def test_template_round_joins_stale_nonfinite_and_meshes_too_large(rt, wl, oracle, monkeypatch):
    """The error paths of a Round-join template: a changed record / a non-finite transform in ANY instance (also one whose meshes only
    the sizes pass reads first) -> VGX_E_STALE / VGX_E_NONFINITE (the caller discards the buffers); a transform under which a mesh outgrows its 16-bit
    indices -> VGX_E_MESH_TOO_LARGE, the status the ordinary pipeline gives for the same batch."""
    ps = wl.closed_fuzz_paths(7995, npaths=72)
    d = wl.template_general_draws(ps, 7995, 40, round_joins=True)
    n = d.shape[0]
    ctx = rt.Context(0)
    for field, where, value in (("stroke_width", n - 1, np.float32(2.5)), ("scale", n // 2, np.float32(1.25)), ("stroke_flags", 17 * ps.npaths + 3, np.uint32(0))):
        d3 = d.copy()
        if d3[field][where] == value:
            value = value + 1
        d3[field][where] = value
        got = _run(rt, ctx, ps, d, d_steady=d3)
        assert got.mode == MODE_TEMPLATE and got.status == VGX_E_STALE, (field, got.status)
    stroked = np.flatnonzero((d["stroke_flags"] & 1) != 0)
    d4 = d.copy()
    d4["mtx"][stroked[-1], 0] = np.float32("inf")
    got = _run(rt, ctx, ps, d, d_steady=d4)
    assert got.status == 3  # VGX_E_NONFINITE
    ctx.close()
    # a mesh that outgrows its 16-bit indices in ONE instance only: an 8000-gon stroked so wide that every join's arc has the minimum two
    # segments (64 000 vertices: fits) -- until an instance squeezes it flat and the turning concentrates in a few dozen joins at both ends
    pm = __import__("importlib").import_module("vg-renderer_amd.pathset")
    b = pm.PathSetBuilder()
    b.begin_path()
    N = 8000
    ang = np.arange(N) * (2.0 * np.pi / N)
    b.move_to(100.0, 0.0)
    for k in range(1, N):
        b.line_to(float(100.0 * np.cos(ang[k])), float(100.0 * np.sin(ang[k])))
    b.close()
    b.end_path()
    b.begin_path()
    b.rect(0.0, 0.0, 10.0, 10.0)
    b.end_path()
    ps2 = b.arrays()
    K = 1100
    one = wl.make_draws(2)
    one["path"] = np.arange(2, dtype=np.uint32)
    one["tess_tol"] = np.float32(1.0e-5)
    wl.set_stroke(one, 0, 0xFF00FF00, 100.0, rt.capi.CAP_BUTT, rt.capi.JOIN_ROUND, aa=True)
    wl.set_fill(one, 1, 0xFF0000FF, aa=True)
    d = np.tile(one, K)
    d5 = d.copy()
    d5["mtx"][-2:, 3] = np.float32(0.01)
    ctx = rt.Context(0)
    ok = _run(rt, ctx, ps2, d, shrink=1.0)
    assert ok.mode == MODE_TEMPLATE and ok.status == 0 and ok.sizes["num_vertices"] == K * (8 * N + 8)
    got = _run(rt, ctx, ps2, d, d_steady=d5, shrink=1.0)
    ctx.close()
    monkeypatch.setenv("VGX_TMPL_ROUND", "0")
    ctx = rt.Context(0)
    pset = rt.PathSet(ctx, ps2)
    dd = rt.upload_draws(d5)
    try:
        rt.tessellate_count(ctx, pset, dd, d5.shape[0])
        ordinary = 0
    except rt.VgxError as e:
        ordinary = e.status
    pset.close()
    ctx.close()
    assert ordinary != 0 and got.status == ordinary, (got.status, ordinary)  # VGX_E_MESH_TOO_LARGE on both
    # the same through the other shape of the sizes pass (ONE scan over all meshes: a static batch of 80 draws)
    monkeypatch.delenv("VGX_TMPL_ROUND")
    ctx = rt.Context(0)
    ctx.set_static_batches(True)
    got2 = _run(rt, ctx, ps2, d[:80], d_steady=d5[-80:], shrink=1.0)
    assert got2.mode == MODE_TEMPLATE and got2.status == ordinary, (got2.mode, got2.status, ordinary)
    ctx.close()


@pytest.mark.parametrize("seed,ninst,max_vb,split", [(6995, 40, 65536, False), (6996, 36, 2048, True), (6997, 50, 700, True), (6998, 70, 65536, True), (6999, 96, 3000, False)])
def test_template_round_joins_with_draw_command_assembly(rt, wl, oracle, monkeypatch, seed, ninst, max_vb, split):
    """Round-join templates with draw-command assembly armed: the assembly's partition reads this step's mesh table (k_tmpl_mtab from the
    per-step places). Vertex / index buffers and draw commands == the ordinary pipeline's (VGX_TMPL_ROUND=0) byte for byte."""
    ps = wl.closed_fuzz_paths(seed, npaths=72)
    d = wl.template_general_draws(ps, seed, ninst, round_joins=True)
    d["state_key"] = (np.arange(d.shape[0]) // 37).astype(d["state_key"].dtype)
    ctx = rt.Context(0)
    got = _assembled(rt, ctx, ps, d, max_vb, split)
    assert got.mode == MODE_TEMPLATE
    assert got.stages[:2] == ["tmpl_round_sizes", "tmpl_mesh_table"], got.stages
    ctx.close()
    monkeypatch.setenv("VGX_TMPL_ROUND", "0")
    ctx = rt.Context(0)
    old = _assembled(rt, ctx, ps, d, max_vb, split)
    assert old.mode != MODE_TEMPLATE
    assert got.status == old.status, (got.status, old.status)  # (700-vertex buffers: a Round-join mesh of a wide stroke does not fit one -- the same error either way)
    if old.status != 0:
        ctx.close()
        return
    assert got.stages[-1] == "tmpl_emit" and got.ncmd == old.ncmd
    for k in ("pos", "color", "idx", "meshes", "cmds"):
        assert bytes_equal(getattr(got, k), getattr(old, k)), k
    ctx.close()


@pytest.mark.parametrize("seed,ninst,ncls,max_vb,split", [(7101, 70, 3, 65536, False), (7102, 96, 6, 2048, True), (7103, 130, 12, 3000, True)])
def test_template_round_joins_in_several_classes_with_draw_command_assembly(rt, wl, oracle, monkeypatch, seed, ninst, ncls, max_vb, split):
    """The same with several classes: k_tmpl_mtab finds a mesh's instance through VgxTmplInst::m and its place in the per-step table."""
    ps = wl.closed_fuzz_paths(seed, npaths=72)
    d, pick = wl.template_class_round_draws(ps, seed, ninst, ncls, closed_aa_only=bool(seed & 1))
    d["state_key"] = (np.arange(d.shape[0]) // 37).astype(d["state_key"].dtype)
    ctx = rt.Context(0)
    got = _assembled(rt, ctx, ps, d, max_vb, split)
    assert got.mode == MODE_TEMPLATE
    assert got.stages[:2] == ["tmpl_round_sizes", "tmpl_mesh_table"], got.stages
    ctx.close()
    monkeypatch.setenv("VGX_TMPL_ROUND", "0")
    ctx = rt.Context(0)
    old = _assembled(rt, ctx, ps, d, max_vb, split)
    assert old.mode != MODE_TEMPLATE
    assert got.status == old.status, (got.status, old.status)
    if old.status != 0:
        ctx.close()
        return
    assert got.stages[-1] == "tmpl_emit" and got.ncmd == old.ncmd
    for k in ("pos", "color", "idx", "meshes", "cmds"):
        assert bytes_equal(getattr(got, k), getattr(old, k)), k
    ctx.close()


@pytest.mark.parametrize("seed,ninst,tile", [(996, 40, None), (997, 36, "192"), (998, 50, "64")])
def test_template_open_miter_strokes(rt, wl, oracle, monkeypatch, seed, ninst, tile):
    """The commonest open style -- Miter joins with Butt or Square caps, AA or hairline -- has an element routine of its own in
    the general kernel (fixed sizes, like the closed one): open and closed fuzz paths, half of the strokes with Square caps."""
    if tile:
        monkeypatch.setenv("VGX_TMPL_TILE", tile)
    ps = wl.fuzz_paths(seed, npaths=72, with_shapes=True, degenerate=False)
    d = wl.template_draws(ps, seed, ninst)
    sq = (np.arange(d.shape[0]) % ps.npaths) % 2 == 1
    stroked = (d["stroke_flags"] & 1) != 0
    d["stroke_flags"][sq & stroked] |= np.uint32(rt.capi.CAP_SQUARE << 4)
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.status == 0
    assert_mesh_equal(got, ref, "template open Miter strokes seed=%d" % seed)
    ctx.close()



def test_template_meshless_draws_between_tiles_are_verified(rt, wl, oracle, monkeypatch):
    """a draw WITHOUT a mesh (fill and stroke disabled) that sits between the last mesh of one tile and the first mesh of
    the next one must still be read by some workgroup -- enabling its fill after the count ends the step with VGX_E_STALE, never
    with a frame in which that mesh is silently missing. Small tiles, every draw position tried."""
    monkeypatch.setenv("VGX_TMPL_TILE", "64")
    ps = wl.closed_fuzz_paths(975, npaths=48)
    NI = 64                                           # (a template needs more than 2048 draws)
    d = wl.template_draws(ps, 975, NI)
    P = ps.npaths
    one = d[:P].copy()
    off = np.arange(P) % 3 == 1                       # every third draw of the period: no mesh at all
    d = d.copy()
    for k in range(NI):
        d["fill_flags"][k * P:(k + 1) * P][off] = 0
        d["stroke_flags"][k * P:(k + 1) * P][off] = 0
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.status == 0
    assert_mesh_equal(got, oracle.tessellate(ps, d), "mesh-less draws")
    for j in np.flatnonzero(off):
        d3 = d.copy()
        d3["fill_flags"][17 * P + j] = one["fill_flags"][j] | 1  # the draw gets its fill back in ONE instance
        assert _run(rt, ctx, ps, d, d_steady=d3).status == VGX_E_STALE, int(j)
    ctx.close()


def test_template_does_not_survive_its_path_set(rt, wl, oracle):
    """a path set destroyed and another one created (possibly at the same address) between the count and the step: the
    template of the old set must not be used for draws that merely look the same."""
    import torch
    psA = wl.closed_fuzz_paths(981, npaths=40)
    psB = wl.closed_fuzz_paths(982, npaths=40)  # same number of paths, other geometry
    d = wl.template_draws(psA, 981, 64)  # (a template needs more than 2048 draws)
    ctx = rt.Context(0)
    for _ in range(6):  # several rounds: the allocator is free to hand the old address out again
        pa = rt.PathSet(ctx, psA)
        dd = rt.upload_draws(d)
        sizes = rt.tessellate_count(ctx, pa, dd, d.shape[0])
        assert ctx.failure_info()["segment_items"] == MODE_TEMPLATE
        pa.close()
        pb = rt.PathSet(ctx, psB)
        ref = oracle.tessellate(psB, d)
        bufs = rt.MeshBuffers(dd.device, max(sizes["num_vertices"], ref.sizes["num_vertices"]), max(sizes["num_indices"], ref.sizes["num_indices"]),
                              max(sizes["num_meshes"], ref.sizes["num_meshes"]))
        try:
            rt.tessellate_async(ctx, pb, dd, d.shape[0], bufs)
            torch.cuda.synchronize()
            st = int(bufs.dev_status.item())
        except rt.VgxError as e:  # "run vgx_tessellate_count once": also fine, nothing was written
            st = e.status
        if st == 0:  # the ordinary pipeline ran on the NEW set: then the output is the new set's
            nv = ref.sizes["num_vertices"]
            assert int(bufs.dev_sizes.cpu().numpy()[3]) == nv
            assert np.array_equal(bufs.pos[:nv].cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
        pb.close()
    ctx.close()


# ---- static batches: the whole draw list as one template (vgx_set_static_batches) ---------------------------------------------------
def _scene(wl, ninst, seed, round_joins=False, keep=0.7):
    """An instanced scene after culling and reordering: `ninst` tigers, a random `keep` of their draws, shuffled -- no period left."""
    ps, ops = wl.tiger_paths()
    d = wl.tiger_draws(ops, ninst, join=1 if round_joins else 0)
    rs = np.random.RandomState(seed)
    d = d[rs.uniform(size=d.shape[0]) < keep]
    return ps, d[rs.permutation(d.shape[0])]


@pytest.mark.parametrize("seed,round_joins,tile", [(11, False, None), (12, False, "192"), (13, True, None), (14, True, "64")])
def test_static_batches_scene_without_a_period(rt, wl, oracle, monkeypatch, seed, round_joins, tile):
    """A culled, shuffled instanced scene has no period: by default the ordinary pipeline (k_flatten_inst grouped by path). With
    vgx_set_static_batches the count flattens the draw list once, in local space, and keeps it as ONE template of one instance; a
    step is then the emit kernel alone (with Round joins: the per-step sizes in front of it). == the reference, == the ordinary
    pipeline byte for byte; new transforms and colours are free, a structural change is VGX_E_STALE."""
    if tile:
        monkeypatch.setenv("VGX_TMPL_TILE", tile)
    ps, d = _scene(wl, 24, seed, round_joins)
    assert d.shape[0] > 2048
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    old = _run(rt, ctx, ps, d)
    assert old.mode != MODE_TEMPLATE and old.status == 0
    ctx.set_static_batches(True)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == (ROUND_STAGES if round_joins else ["tmpl_emit"]), (got.mode, got.stages)
    assert got.status == 0
    assert_mesh_equal(got, ref, "static batch seed=%d" % seed)
    for k in ("pos", "color", "idx", "meshes"):
        assert bytes_equal(getattr(got, k), getattr(old, k)), k
    two = _run(rt, ctx, ps, d, two_phase=True)  # vgx_tessellate_count + vgx_tessellate_emit: the same template, the same bytes
    assert two.mode == MODE_TEMPLATE
    for k in ("pos", "color", "idx", "meshes"):
        assert bytes_equal(getattr(two, k), getattr(old, k)), k
    # the camera moves, colours change: the same template
    d2 = d.copy()
    rs = np.random.RandomState(seed + 100)
    m = rs.uniform(-2, 2, size=6).astype(np.float32)
    d2["mtx"][:] = m
    d2["fill_color"] = rs.randint(0, 1 << 32, size=d.shape[0], dtype=np.uint64).astype(np.uint32)
    if not round_joins:  # (Round joins: the sizes follow the transform -- covered by test_template_round_joins_sizes_follow_the_transforms)
        got2 = _run(rt, ctx, ps, d, d_steady=d2)
        assert got2.mode == MODE_TEMPLATE and got2.status == 0
        assert_mesh_equal(got2, oracle.tessellate(ps, d2), "static batch, new camera")
    # two draws swapped / one stroke width changed: stale
    d3 = d.copy()
    a, b = 5, d.shape[0] - 7
    if d3["path"][a] == d3["path"][b]:
        b -= 1
    d3[[a, b]] = d3[[b, a]]
    assert _run(rt, ctx, ps, d, d_steady=d3).status == VGX_E_STALE
    ctx.set_static_batches(False)
    back = _run(rt, ctx, ps, d)
    assert back.mode != MODE_TEMPLATE and back.status == 0
    ctx.close()


def test_static_batches_every_draw_its_own_path(rt, wl, oracle):
    """No draw shares a path with another (what a whole retained frame looks like): still one template."""
    ps = wl.fuzz_paths(4242, npaths=2600, with_shapes=True, degenerate=False)
    d = wl.template_general_draws(ps, 4242, 1, round_joins=True)
    assert d.shape[0] == 2600
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    ctx.set_static_batches(True)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.status == 0
    assert_mesh_equal(got, ref, "static batch: one frame")
    ctx.close()


def test_static_batches_long_round_join_polylines(rt, wl, oracle):
    """BASELINE configs[3]'s shape (open polylines of a thousand segments, Round joins + Round caps) as a static batch: the per-step
    sizes by one WORKGROUP per mesh (k_tmpl_round_sizes_block), meshes that span several tiles, general Round-join emit."""
    ps, d = wl.random_walk_polylines(2100, 700, seed=99)
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    ctx.set_static_batches(True)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == ROUND_STAGES and got.status == 0
    assert_mesh_equal(got, ref, "static batch: long round-join polylines")
    d2 = d.copy()
    d2["mtx"][:, 0] = np.float32(0.7); d2["mtx"][:, 3] = np.float32(1.3); d2["mtx"][:, 1] = np.float32(0.2)
    ref2 = oracle.tessellate(ps, d2)
    import torch
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    rt.tessellate_count(ctx, pset, dd, d.shape[0])
    dd2 = rt.upload_draws(d2)
    bufs = rt.MeshBuffers(dd2.device, ref2.pos.shape[0] + 64, ref2.idx.shape[0] + 64, int(ref2.meshes.shape[0]))
    rt.tessellate_async(ctx, pset, dd2, d2.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    assert bytes_equal(bufs.pos[:ref2.pos.shape[0]].cpu().numpy(), ref2.pos) and bytes_equal(bufs.idx[:ref2.idx.shape[0]].cpu().numpy().view(np.uint16), ref2.idx)
    pset.close()
    ctx.close()


@pytest.mark.parametrize("max_vb,split", [(65536, False), (3000, True)])
def test_static_batches_one_frame(rt, wl, oracle, max_vb, split):
    """A frame-sized draw list (one tiger, 240 draws; below the 2 048 draws template mode otherwise asks for) as a static batch, meshes
    only and with draw-command assembly armed: the reference's bytes, the ordinary frame path's draw commands."""
    ps, d = wl.tiger(1)
    d["state_key"] = (np.arange(d.shape[0]) // 50).astype(d["state_key"].dtype)
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    old = _assembled(rt, ctx, ps, d, max_vb, split)
    assert old.mode != MODE_TEMPLATE and old.status == 0
    ctx.set_static_batches(True)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == ["tmpl_emit"] and got.status == 0
    assert_mesh_equal(got, ref, "static batch: one frame")
    asm = _assembled(rt, ctx, ps, d, max_vb, split)
    assert asm.mode == MODE_TEMPLATE and asm.status == 0 and "tmpl_emit" in asm.stages
    assert asm.ncmd == old.ncmd
    for k in ("pos", "color", "idx", "meshes", "cmds"):
        assert bytes_equal(getattr(asm, k), getattr(old, k)), k
    ctx.close()


@pytest.mark.parametrize("seed,ninst,tile", [(8101, 40, None), (8102, 36, "128"), (8103, 33, "64")])
def test_template_open_and_closed_aa_strokes_with_round_joins_and_any_cap(rt, wl, oracle, monkeypatch, seed, ninst, tile):
    """Every stroke AA (wider than the fringe) with Round joins, open and closed sub-paths, Butt / Square / Round caps: the template
    kernel without the general body (tmpl_stroke_elem_round + tmpl_stroke_cap_aa beside the Miter / Bevel routines). == the reference,
    == the ordinary pipeline byte for byte."""
    if tile:
        monkeypatch.setenv("VGX_TMPL_TILE", tile)
    ps = wl.fuzz_paths(seed, npaths=72, with_shapes=True, degenerate=False)
    d = wl.template_draws(ps, seed, ninst)
    n = ps.npaths
    rs = np.random.RandomState(seed)
    one = d[:n].copy()
    for i in range(n):
        if one["stroke_flags"][i] & 1:
            wl.set_stroke(one, i, int(rs.randint(0, 1 << 32, dtype=np.uint64)), float(rs.choice([2.5, 3.0, 6.0, 12.0])),
                          int(rs.choice([rt.capi.CAP_BUTT, rt.capi.CAP_SQUARE, rt.capi.CAP_ROUND])), rt.capi.JOIN_ROUND, aa=True,
                          avg_scale=float(one["scale"][i]), fringe=float(one["fringe"][i]))
    for k in ("stroke_flags", "stroke_width", "stroke_color"):
        d[k] = np.tile(one[k], ninst)
    ref = oracle.tessellate(ps, d)
    kinds = np.unique(ref.meshes["subpath_kind"] >> 24) if False else None
    ctx = rt.Context(0)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.stages == ROUND_STAGES and got.status == 0
    assert_mesh_equal(got, ref, "round joins, open + closed AA strokes seed=%d" % seed)
    ctx.close()
    monkeypatch.setenv("VGX_TMPL_ROUND", "0")
    ctx = rt.Context(0)
    old = _run(rt, ctx, ps, d)
    assert old.mode != MODE_TEMPLATE
    for k in ("pos", "color", "idx", "meshes"):
        assert bytes_equal(getattr(got, k), getattr(old, k)), k
    ctx.close()


@pytest.mark.parametrize("seed", [9001, 9002, 9003, 9004])
def test_static_batches_fuzz_with_degenerate_paths(rt, wl, oracle, seed):
    """Static batches over the full fuzz grammar -- degenerate steps inside the epsilon ball, sub-paths closing onto their start point,
    every shape command, every stroke style (Round joins among them), draws without fill or stroke -- in a shuffled order: the count's
    flatten takes the exact serial kernel for the degenerate draws, the template holds what it produced."""
    ps = wl.fuzz_paths(seed, npaths=96, with_shapes=True, degenerate=True)
    d = wl.template_general_draws(ps, seed, 30, round_joins=True)
    rs = np.random.RandomState(seed)
    d = d[rs.permutation(d.shape[0])]
    d["mtx"] = rs.uniform(-2.5, 2.5, size=d["mtx"].shape).astype(np.float32)
    ref = oracle.tessellate(ps, d)
    ctx = rt.Context(0)
    ctx.set_static_batches(True)
    got = _run(rt, ctx, ps, d)
    assert got.mode == MODE_TEMPLATE and got.status == 0
    assert_mesh_equal(got, ref, "static batch fuzz seed=%d" % seed)
    ctx.close()


def test_static_batches_beyond_the_size_limit_keep_the_ordinary_pipeline(rt, wl):
    """One template instance addresses its streams with 32-bit offsets (2^29 vertices, 2^31 indices / elements): a static batch beyond that
    is not made a template -- the ordinary pipeline runs, same call sequence, same bytes as without the promise."""
    import torch
    ps, ops = wl.tiger_spec_paths()
    d = wl.tiger_draws(ops, 4200)
    rs = np.random.RandomState(5)
    d = d[rs.uniform(size=d.shape[0]) < 0.7]
    d = d[rs.permutation(d.shape[0])]
    outs = []
    for static in (True, False):
        ctx = rt.Context(0)
        ctx.set_static_batches(static)
        pset = rt.PathSet(ctx, ps)
        dd = rt.upload_draws(d)
        sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
        assert sizes["num_vertices"] > (1 << 29)
        assert ctx.failure_info()["segment_items"] != MODE_TEMPLATE
        bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
        rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
        torch.cuda.synchronize()
        assert int(bufs.dev_status.item()) == 0
        outs.append((sizes, bufs))
        pset.close()
        ctx.close()
    (sa, a), (sb, b) = outs
    assert sa == sb
    nv, ni = sa["num_vertices"], sa["num_indices"]
    assert torch.equal(a.pos[:nv].view(torch.int32), b.pos[:nv].view(torch.int32)) and torch.equal(a.color[:nv], b.color[:nv]) and torch.equal(a.idx[:ni], b.idx[:ni])
