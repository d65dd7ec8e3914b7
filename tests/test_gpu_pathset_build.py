"""vgx_pathset_create builds the path set's derived tables on the DEVICE (round 6, csrc/vgx_pathset.hip): every table of every
set == the host loops it replaced (csrc/vgx_pathset_host.h, csrc/vgx_thin.h -- compiled into libvgx_hosttest.so, the oracle of
these kernels), byte for byte. Invalid sets: the same status the host validator names, whatever the device flagged."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLES = {"cmd_flags": 0, "sp_start": 1, "path_flags": 2, "cmdrec": 3, "path_sub_begin": 4, "sub_last_cmd": 5, "cmdthin": 6, "thin_path": 7, "thin_sub": 8, "scalars": 9}


@pytest.fixture(scope="module")
def rt():
    return importlib.import_module("vg-renderer_amd.runtime")


@pytest.fixture(scope="module")
def hostlib():
    path = os.path.join(ROOT, "vg-renderer_amd", "libvgx_hosttest.so")
    if not os.path.exists(path):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(path)
    lib.vgxt_pathset_table.restype = C.c_int64
    lib.vgxt_pathset_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    return lib


def host_table(hostlib, desc, which):
    n = hostlib.vgxt_pathset_table(C.addressof(desc), which, None, 0)
    assert n >= 0, n
    buf = np.zeros(max(int(n), 1), dtype=np.uint8)
    assert hostlib.vgxt_pathset_table(C.addressof(desc), which, buf.ctypes.data, buf.nbytes) == n
    return buf[:n]


def dev_table(rt, ctx, pset, which):
    L = rt.lib()
    n = C.c_uint64(0)
    rt._check(L.vgx_pathset_read_table(ctx.handle, pset.handle, which, None, 0, C.byref(n)), "vgx_pathset_read_table")
    buf = np.zeros(max(int(n.value), 1), dtype=np.uint8)
    rt._check(L.vgx_pathset_read_table(ctx.handle, pset.handle, which, buf.ctypes.data, buf.nbytes, C.byref(n)), "vgx_pathset_read_table")
    return buf[:n.value]


def check_set(rt, hostlib, ctx, ps, what):
    pset = rt.PathSet(ctx, ps)
    desc = ps.desc()
    scal = dev_table(rt, ctx, pset, TABLES["scalars"]).view(np.uint32)
    href = host_table(hostlib, desc, TABLES["scalars"]).view(np.uint32)
    assert np.array_equal(scal, href), (what, "scalars", scal, href)
    thin_static = bool(scal[3])
    for name, which in TABLES.items():
        if name == "scalars":
            continue
        got = dev_table(rt, ctx, pset, which)
        ref = host_table(hostlib, desc, which)
        if name == "cmdthin" and not thin_static:
            # vgx_thin_build leaves `pad` and the upper half of `meta` of an ineligible set as they were (zero, or half-way through
            # the paths when a path has > 65 536 sub-paths): nobody reads them then. The points and type | flags << 8 are read.
            g, r = got.view(np.uint32).reshape(-1, 4).copy(), ref.view(np.uint32).reshape(-1, 4).copy()
            g[:, 0] &= 0xFFFF
            r[:, 0] &= 0xFFFF
            g[:, 3] = 0
            r[:, 3] = 0
            got, ref = g.view(np.uint8).reshape(-1), r.view(np.uint8).reshape(-1)
        if not np.array_equal(got, ref):
            bad = np.nonzero(got != ref)[0] if got.shape == ref.shape else None
            raise AssertionError((what, name, got.shape, ref.shape, None if bad is None else (int(bad[0]), len(bad))))
    pset.close()
    return scal


def test_fuzz_sets_every_command(rt, hostlib, gpu_ctx, wl):
    for seed in range(40):
        ps = wl.fuzz_paths(seed, npaths=48 + 7 * (seed % 5), with_shapes=True, degenerate=True)
        check_set(rt, hostlib, gpu_ctx, ps, ("fuzz", seed))


def test_fuzz_sets_with_polylines_and_without_shapes(rt, hostlib, gpu_ctx, wl):
    for seed in range(20):
        check_set(rt, hostlib, gpu_ctx, wl.fuzz_paths(100 + seed, npaths=64, with_shapes=False, degenerate=bool(seed & 1), with_polylines=True), ("poly", seed))


def test_thin_sets_get_the_static_layout(rt, hostlib, gpu_ctx, wl):
    saw_static = saw_degenerate = 0
    for seed in range(40):
        ps = wl.thin_fuzz_paths(seed, npaths=32 + seed, degenerate=bool(seed % 3 == 0))
        scal = check_set(rt, hostlib, gpu_ctx, ps, ("thin", seed))
        saw_static += int(scal[3])
        saw_degenerate += int(seed % 3 == 0)
    assert saw_static == 40 and saw_degenerate > 0


def test_closed_fuzz_and_bench_sets(rt, hostlib, gpu_ctx, wl):
    for seed in range(8):
        check_set(rt, hostlib, gpu_ctx, wl.closed_fuzz_paths(seed), ("closed", seed))
    check_set(rt, hostlib, gpu_ctx, wl.tiger_paths()[0], "tiger")
    check_set(rt, hostlib, gpu_ctx, wl.tiger_paths(closed=False)[0], "tiger open")
    check_set(rt, hostlib, gpu_ctx, wl.tiger_spec_paths()[0], "tigerspec")
    check_set(rt, hostlib, gpu_ctx, wl.single_cubic()[0], "single cubic")


def test_sets_longer_than_one_scan_slice(rt, hostlib, gpu_ctx, wl):
    """Carries across the scan's slices and tiles: 300 000 cubics (600 000 commands: ~1 200 per slice), 2 000 polylines x 300
    segments (a path spans several tiles; closing / popping pathClose at the end of some), and empty paths in between."""
    ps, _ = wl.random_cubics(300000, seed=9)
    check_set(rt, hostlib, gpu_ctx, ps, "cubics 300k")
    ps, _ = wl.random_walk_polylines(2000, 300, seed=11)
    scal = check_set(rt, hostlib, gpu_ctx, ps, "polylines 2000 x 300")
    assert scal[3] == 1 and scal[0] == 301
    # closed polygons whose last point repeats the first (pathClose pops it), open ones, two-point ones, empty paths
    b = importlib.import_module("vg-renderer_amd").PathSetBuilder()
    rs = np.random.RandomState(5)
    for k in range(3000):
        b.begin_path()
        if k % 7 != 3:
            for sub in range(1 + k % 3):
                n = int(rs.randint(1, 40))
                pts = rs.uniform(0, 100, size=(n, 2)).astype(np.float32)
                b.move_to(float(pts[0, 0]), float(pts[0, 1]))
                for q in pts[1:]:
                    b.line_to(float(q[0]), float(q[1]))
                if k % 2:
                    if k % 4 == 1 and n > 3:
                        b.line_to(float(pts[0, 0]), float(pts[0, 1]))  # the vertex pathClose removes (path.cpp:716-725)
                    b.close()
        b.end_path()
    ps = b.arrays()
    scal = check_set(rt, hostlib, gpu_ctx, ps, "polygons with empty paths")
    assert scal[2] == 1 and scal[3] == 0  # empty paths: no static layout, but every thin path keeps VGX_PF_THIN
    # the same without the empty paths: static layout
    b = importlib.import_module("vg-renderer_amd").PathSetBuilder()
    for k in range(3000):
        b.begin_path()
        for sub in range(1 + k % 3):
            n = int(rs.randint(1, 40))
            pts = rs.uniform(0, 100, size=(n, 2)).astype(np.float32)
            b.move_to(float(pts[0, 0]), float(pts[0, 1]))
            for q in pts[1:]:
                b.line_to(float(q[0]), float(q[1]))
            if k % 2:
                if k % 4 == 1 and n > 3:
                    b.line_to(float(pts[0, 0]), float(pts[0, 1]))
                b.close()
        b.end_path()
    scal = check_set(rt, hostlib, gpu_ctx, b.arrays(), "polygons")
    assert scal[3] == 1


def test_frame_sized_sets_through_the_large_set_sequence(rt, hostlib, wl, monkeypatch):
    """Frame-sized sets go up as one image into a recycled blob and skip the thin passes when the host sees a curve among their opcodes;
    VGX_PS_NO_SMALL sends them the large sets' way (four uploads, fresh blob, every pass launched)."""
    monkeypatch.setenv("VGX_PS_NO_SMALL", "1")
    ctx = rt.Context(0)
    monkeypatch.delenv("VGX_PS_NO_SMALL")
    for seed in range(12):
        check_set(rt, hostlib, ctx, wl.fuzz_paths(300 + seed, npaths=64, with_shapes=True, degenerate=True), ("fuzz, large sequence", seed))
        check_set(rt, hostlib, ctx, wl.thin_fuzz_paths(300 + seed, npaths=40, degenerate=bool(seed & 1)), ("thin, large sequence", seed))
    ctx.close()


def test_dropped_sets_are_recycled(rt, hostlib, gpu_ctx, wl):
    """Blobs of dropped frame-sized sets wait in the context for the next create: sets of different sizes made and dropped in turn keep
    giving the host loops' tables (stale bytes of an earlier, larger set must not show)."""
    for rnd in range(3):
        for seed in (5, 1, 9, 3):
            check_set(rt, hostlib, gpu_ctx, wl.fuzz_paths(seed, npaths=16 + 24 * seed), ("recycled", rnd, seed))
            check_set(rt, hostlib, gpu_ctx, wl.thin_fuzz_paths(seed, npaths=8 + 9 * seed), ("recycled thin", rnd, seed))


def test_empty_and_tiny_sets(rt, hostlib, gpu_ctx, vgr):
    b = vgr.PathSetBuilder()
    check_set(rt, hostlib, gpu_ctx, b.arrays(), "no paths")
    b = vgr.PathSetBuilder()
    b.begin_path()
    b.end_path()
    check_set(rt, hostlib, gpu_ctx, b.arrays(), "one empty path")
    b = vgr.PathSetBuilder()
    b.begin_path()
    b.move_to(1, 2)
    b.end_path()
    check_set(rt, hostlib, gpu_ctx, b.arrays(), "one moveTo")


def test_invalid_sets_get_the_validators_status(rt, gpu_ctx, vgr, wl):
    capi = vgr.capi

    def create_status(arrays):
        h = C.c_void_p()
        d = arrays.desc()
        st = rt.lib().vgx_pathset_create(gpu_ctx.handle, C.byref(d), C.byref(h))
        if st == capi.VGX_OK:
            rt.lib().vgx_pathset_destroy(gpu_ctx.handle, h)
        assert st == rt.validate_pathset(arrays), (st, rt.validate_pathset(arrays))
        return st

    def one(build):
        b = vgr.PathSetBuilder()
        b.begin_path()
        build(b)
        b.end_path()
        return create_status(b.arrays())

    assert one(lambda b: (b.move_to(0, 0), b.line_to(1, 1), b.close())) == capi.VGX_OK
    assert one(lambda b: b.line_to(1, 1)) == capi.VGX_E_INVALID_PATH
    assert one(lambda b: (b.move_to(0, 0), b.line_to(1, 0), b.line_to(1, 1), b.close(), b.line_to(2, 2))) == capi.VGX_E_INVALID_PATH
    assert one(lambda b: (b.rect(0, 0, 1, 1), b.line_to(2, 2))) == capi.VGX_E_INVALID_PATH
    assert one(lambda b: (b.move_to(0, 0), b.close(), b.arc(0, 0, 5, 0, 1, True))) == capi.VGX_E_INVALID_PATH
    assert one(lambda b: (b.arc(0, 0, 5, 0, 2.0e5, True),)) == capi.VGX_E_INVALID_ARG
    assert one(lambda b: (b.move_to(0, 0), b.line_to(float("nan"), 1))) == capi.VGX_E_NONFINITE
    assert one(lambda b: (b.move_to(0, 0), b.cubic_to(1, 1, 2, float("inf"), 3, 3))) == capi.VGX_E_NONFINITE
    # an error deep inside a large valid set (the flag must survive the scan's slices); a bad opcode; a bad argument count
    ps, _ = wl.random_cubics(200000, seed=3)
    bad = vgr.pathset.PathSetArrays(ps.cmd_type.copy(), ps.cmd_arg_off.copy(), ps.args.copy(), ps.path_cmd_begin.copy())
    bad.cmd_type[250001] = capi.CMD_LINE_TO  # cubicTo -> lineTo: six arguments for a lineTo
    assert create_status(bad) == capi.VGX_E_INVALID_ARG
    bad.cmd_type[250001] = 77
    assert create_status(bad) == capi.VGX_E_INVALID_ARG
    bad.cmd_type[250001] = capi.CMD_CUBIC_TO
    assert create_status(bad) == capi.VGX_OK
    bad.cmd_type[123456] = capi.CMD_CUBIC_TO  # a path that begins with cubicTo (its moveTo replaced; argument count wrong too)
    assert create_status(bad) in (capi.VGX_E_INVALID_ARG, capi.VGX_E_INVALID_PATH)
    bad.cmd_type[123456] = capi.CMD_MOVE_TO
    bad.args[777777] = np.float32("nan")
    assert create_status(bad) == capi.VGX_E_NONFINITE
    bad.args[777777] = 1.0
    bad.path_cmd_begin[100] = bad.path_cmd_begin[101] + 1  # not monotone
    assert create_status(bad) == capi.VGX_E_INVALID_ARG
