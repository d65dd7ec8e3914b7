"""k_emit_tiles (csrc/vgx_tile.hip, round 6): the draw-ordered tile kernel that emits the fills and closed Miter AA / Thin strokes of
ORDINARY batches (no template, structure free to change every call) in place of k_fill + k_stroke_simple. It is launched for calls with
room for >= 2^18 vertices whose last count found no other stroke style, so every large ordinary-pipeline test of the suite runs through it
(tests/test_gpu_golden.py full-size comparisons, tests/test_gpu_inst.py); this file adds what is specific to it: fuzz drawings with every
fill flavour (AA / plain / SSE index order / none) and hairline + regular closed strokes under arbitrary transforms, with and without the
instanced flattener in front, shuffled draw order (meshes of every size next to each other), tiles with more meshes than the LDS tables
hold (the per-lane fallback), draw-command assembly armed (index bases), the two-phase entry, and the hand-over to k_fill + k_stroke when
the batch turns out to hold another stroke style. Bit-exact against the reference."""
import importlib
import os

import numpy as np
import pytest

from util import assert_mesh_equal, run_async

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


@pytest.mark.parametrize("seed,inst", [(11, 1), (12, 0), (13, 1), (14, 0)])
def test_fuzz_drawings_through_the_tile_kernel(rt, wl, oracle, seed, inst):
    ps = wl.closed_fuzz_paths(seed, npaths=56)
    d = wl.template_draws(ps, seed, 170)
    if seed & 1:
        d = d[np.random.RandomState(seed).permutation(d.shape[0])]  # no period, meshes of every size next to each other
    ctx = _ctx_with(rt, VGX_TMPL=0, VGX_INST=inst)
    got = run_async(rt, ctx, ps, d, profile=True)
    ref = oracle.tessellate(ps, d)
    assert ref.pos.shape[0] > (1 << 18)
    assert got.status == 0 and "tile_emit" in got.stages, got.stages
    assert_mesh_equal(got, ref, "tile kernel, fuzz drawing %d" % seed)
    # the same call with the tile kernel off: the two emit paths agree with each other too
    ctx2 = _ctx_with(rt, VGX_TMPL=0, VGX_INST=inst, VGX_TILE_EMIT=0)
    got2 = run_async(rt, ctx2, ps, d, profile=True)
    assert "tile_emit" not in got2.stages
    assert_mesh_equal(got2, ref, "k_fill + k_stroke_simple, fuzz drawing %d" % seed)
    ctx.close()
    ctx2.close()


def test_tiles_with_more_meshes_than_the_lds_tables_hold(rt, wl, oracle, vgr):
    """Triangles and two-segment closed strokes by the hundred thousand: ~400 meshes per 2 048-element tile (the tables hold 192): the
    per-lane fallback of k_emit_tiles; mixed with a few long polygons so that some tiles take the staged path."""
    b = vgr.PathSetBuilder()
    rs = np.random.RandomState(5)
    for k in range(64):
        b.begin_path()
        n = 3 if k % 8 else 200
        ang = np.sort(rs.uniform(0, 2 * np.pi, size=n))
        r = rs.uniform(3, 9)
        b.move_to(float(r * np.cos(ang[0])), float(r * np.sin(ang[0])))
        for a in ang[1:]:
            b.line_to(float(r * np.cos(a)), float(r * np.sin(a)))
        b.close()
        b.end_path()
    ps = b.arrays()
    n = 60000
    d = vgr.make_draws(n)
    d["path"] = rs.randint(0, 64, size=n)
    wl.set_fill(d, slice(None), 0xFF336699, aa=True)
    sel = np.flatnonzero(rs.uniform(size=n) < 0.5)
    wl.set_stroke(d, sel, 0xFFCC3311, 1.5, rt.capi.CAP_BUTT, rt.capi.JOIN_MITER, aa=True)
    d["mtx"][:, 4] = rs.uniform(0, 2000, size=n).astype(np.float32)
    d["mtx"][:, 5] = rs.uniform(0, 2000, size=n).astype(np.float32)
    ctx = _ctx_with(rt, VGX_TMPL=0, VGX_INST=0)
    got = run_async(rt, ctx, ps, d, profile=True)
    ref = oracle.tessellate(ps, d)
    assert ref.pos.shape[0] > (1 << 18) and "tile_emit" in got.stages
    assert_mesh_equal(got, ref, "tiny meshes")
    ctx.close()


def test_tile_kernel_with_assembly_armed_and_two_phase_entry(rt, wl, oracle):
    import torch
    ps = wl.closed_fuzz_paths(21, npaths=48)
    d = wl.template_draws(ps, 21, 200)
    ref = oracle.tessellate(ps, d)
    max_vb = 4096
    st, rcmds, ridx = oracle.assemble(ref.meshes, ref.idx, max_vb)
    assert st == 0 and len(rcmds) > 50
    ctx = _ctx_with(rt, VGX_TMPL=0, VGX_INST=0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    for use_async in (True, False):
        bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
        cmds = torch.zeros((2 * (nv // max_vb) + 2) * 48, dtype=torch.uint8, device=dd.device)
        ncmd = torch.zeros(1, dtype=torch.int64, device=dd.device)
        ctx.set_assembly(cmds, max_vb, ncmd)
        ctx.set_profiling(True)
        try:
            if use_async:
                rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
            else:
                rt.tessellate_count(ctx, pset, dd, d.shape[0])
                rt.tessellate_emit(ctx, pset, dd, d.shape[0], bufs)
            torch.cuda.synchronize()
            stages = [n for n, _ in ctx.stage_times()]
        finally:
            ctx.set_assembly(None)
            ctx.set_profiling(False)
        assert "tile_emit" in stages and "assemble" in stages, stages
        assert int(ncmd.item()) == len(rcmds)
        assert np.array_equal(bufs.idx[:ni].cpu().numpy().view(np.uint16), ridx), "command-relative indices"
        assert np.array_equal(bufs.pos[:nv].cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
        assert np.array_equal(bufs.color[:nv].cpu().numpy().view(np.uint32), ref.color)
        got_cmds = cmds[:len(rcmds) * 48].cpu().numpy().view(rt.capi.drawcmd_dtype)
        for f in rcmds.dtype.names:
            assert np.array_equal(got_cmds[f], rcmds[f]), f
    pset.close()
    ctx.close()


def test_batch_that_turns_general_goes_back_to_k_fill_and_k_stroke(rt, wl, oracle):
    """The count sees fills + closed Miter strokes (tile kernel armed); the steady-state call then brings draws with Bevel joins and open
    sub-paths: the scan over the meshes finds them on the device, k_emit_tiles exits, k_fill + k_stroke emit the batch -- same bytes as the
    reference either way."""
    import torch
    ps = wl.closed_fuzz_paths(31, npaths=48)
    d = wl.template_draws(ps, 31, 170)
    ctx = _ctx_with(rt, VGX_TMPL=0, VGX_INST=0)
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    rt.tessellate_count(ctx, pset, dd, d.shape[0])
    d2 = d.copy()
    stroked = np.flatnonzero((d2["stroke_flags"] & 1) != 0)
    d2["stroke_flags"][stroked[::3]] = rt.capi.stroke_flags(rt.capi.CAP_BUTT, rt.capi.JOIN_BEVEL, aa=True)
    ref = oracle.tessellate(ps, d2)
    dd2 = rt.upload_draws(d2)
    bufs = rt.MeshBuffers(dd.device, ref.pos.shape[0], ref.idx.shape[0], ref.meshes.shape[0])
    rt.tessellate_async(ctx, pset, dd2, d2.shape[0], bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    assert np.array_equal(bufs.idx[:ref.idx.shape[0]].cpu().numpy().view(np.uint16), ref.idx)
    assert np.array_equal(bufs.pos[:ref.pos.shape[0]].cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
    assert np.array_equal(bufs.color[:ref.pos.shape[0]].cpu().numpy().view(np.uint32), ref.color)
    pset.close()
    ctx.close()
