"""AA fringes of concave fills. The device does the stroker's own loops of strokerConcaveFillEndAA
(reference src/stroker.cpp:868-1006: fringe vertices / indices per boundary contour, contour vertices moved to the inner
fringe vertex, interior appended with rebased indices); libtess2 stays on the CPU side of the caller -- here the test
plays the caller, with the libtess2 that is linked into oracle/_ref/libvgref.so. Checker: the reference's own
strokerConcaveFillBegin / AddContour / EndAA from the same library. Parity pinned: bit-exact positions, colours, indices."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    return importlib.import_module("vg-renderer_amd.runtime")


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.available("reference"):
        pytest.skip("oracle/_ref/libvgref.so (the reference's sources + libtess2) is not built")
    return load_ref(oracle)


def load_ref(oracle):
    lib = oracle.load("reference")
    lib.vgo_concave_fill_aa.restype = C.c_int
    lib.vgo_concave_fill_aa.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.vgo_tess_new.restype = C.c_void_p
    lib.vgo_tess_delete.argtypes = [C.c_void_p]
    lib.vgo_tess_add_contour.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.vgo_tess_run.restype = C.c_int
    lib.vgo_tess_run.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.vgo_tess_vertex_count.restype = C.c_int
    lib.vgo_tess_vertex_count.argtypes = [C.c_void_p]
    lib.vgo_tess_vertices.restype = C.POINTER(C.c_float)
    lib.vgo_tess_vertices.argtypes = [C.c_void_p]
    lib.vgo_tess_element_count.restype = C.c_int
    lib.vgo_tess_element_count.argtypes = [C.c_void_p]
    lib.vgo_tess_elements.restype = C.POINTER(C.c_uint16)
    lib.vgo_tess_elements.argtypes = [C.c_void_p]
    return lib


def _ring(cx, cy, r, n, phase=0.0, cw=False, wobble=0.0, seed=0):
    rs = np.random.RandomState(seed)
    a = phase + np.arange(n) * (2 * np.pi / n) * (-1 if cw else 1)
    rad = r * (1.0 + wobble * rs.uniform(-1, 1, size=n))
    return np.stack([cx + rad * np.cos(a), cy + rad * np.sin(a)], axis=1).astype(np.float32)


def _star(cx, cy, r0, r1, points, phase=0.0):
    a = phase + np.arange(2 * points) * (np.pi / points)
    rad = np.where(np.arange(2 * points) % 2 == 0, r0, r1)
    return np.stack([cx + rad * np.cos(a), cy + rad * np.sin(a)], axis=1).astype(np.float32)


def _fills():
    """(contours, colour, fringe, evenOdd) per concave fill."""
    out = []
    out.append(([_star(100, 100, 80, 30, 5)], 0xFF3366CC, 1.0, 0))                       # concave star
    out.append(([_ring(300, 120, 90, 24), _ring(300, 120, 40, 16, cw=True)], 0x80FF8040, 1.0, 0))  # donut: hole = opposite winding
    out.append(([_ring(60, 300, 50, 12), _ring(110, 300, 50, 12)], 0xFF00AA55, 1.0, 0))  # two overlapping discs, NonZero: union
    out.append(([_ring(60, 300, 50, 12), _ring(110, 300, 50, 12)], 0xFF00AA55, 0.5, 1))  # same, EvenOdd: lens removed
    pent = _star(400, 400, 90, 90, 5)[::2]
    out.append(([pent[[0, 2, 4, 1, 3]]], 0xFFFFFFFF, 1.0, 0))                             # self-intersecting pentagram, NonZero
    out.append(([pent[[0, 2, 4, 1, 3]]], 0xC0102030, 2.0, 1))                             # ... EvenOdd (hollow centre)
    sq = np.array([[0, 0], [40, 0], [40, 40], [0, 40]], dtype=np.float32)
    out.append(([sq + 500, sq + np.float32([540, 540])], 0xFF777777, 1.0, 0))             # two squares touching at a corner
    out.append(([_ring(600, 200, 70, 40, wobble=0.35, seed=3)], 0xFFABCDEF, 1.0, 0))      # wobbly concave blob
    out.append(([_ring(700, 500, 60, 9, cw=True)], 0xFF123456, 1.0, 0))                   # clockwise input
    return out


def _tess_boundary(ref, contours, even_odd):
    t = ref.vgo_tess_new()
    for c in contours:
        c = np.ascontiguousarray(c, dtype=np.float32)
        ref.vgo_tess_add_contour(t, c.ctypes.data, c.shape[0])
    assert ref.vgo_tess_run(t, even_odd, 1) == 1
    nv = ref.vgo_tess_vertex_count(t)
    ne = ref.vgo_tess_element_count(t)
    verts = np.ctypeslib.as_array(ref.vgo_tess_vertices(t), shape=(nv, 2)).copy() if nv else np.zeros((0, 2), np.float32)
    el = np.ctypeslib.as_array(ref.vgo_tess_elements(t), shape=(ne, 2)).copy() if ne else np.zeros((0, 2), np.uint16)
    return t, verts, el


def _tess_polygons(ref, t, moved, el, even_odd):
    for first, n in el:
        seg = np.ascontiguousarray(moved[int(first):int(first) + int(n)])
        ref.vgo_tess_add_contour(t, seg.ctypes.data, int(n))
    assert ref.vgo_tess_run(t, even_odd, 0) == 1
    nv = ref.vgo_tess_vertex_count(t)
    ne = ref.vgo_tess_element_count(t)
    verts = np.ctypeslib.as_array(ref.vgo_tess_vertices(t), shape=(nv, 2)).copy() if nv else np.zeros((0, 2), np.float32)
    idx = np.ctypeslib.as_array(ref.vgo_tess_elements(t), shape=(ne * 3,)).copy() if ne else np.zeros((0,), np.uint16)
    ref.vgo_tess_delete(t)
    return verts, idx


def _reference_mesh(ref, contours, color, fringe, even_odd):
    allv = np.concatenate(contours).astype(np.float32)
    first = np.cumsum([0] + [c.shape[0] for c in contours[:-1]]).astype(np.uint32)
    count = np.array([c.shape[0] for c in contours], dtype=np.uint32)
    cap_v, cap_i = 65536, 65536 * 6
    pos = np.zeros((cap_v, 2), np.float32)
    col = np.zeros(cap_v, np.uint32)
    idx = np.zeros(cap_i, np.uint16)
    nv, ni = C.c_uint32(0), C.c_uint32(0)
    rc = ref.vgo_concave_fill_aa(allv.ctypes.data, first.ctypes.data, count.ctypes.data, len(contours), color, fringe, even_odd,
                                 pos.ctypes.data, col.ctypes.data, idx.ctypes.data, cap_v, cap_i, C.addressof(nv), C.addressof(ni))
    assert rc == 0
    return pos[:nv.value].copy(), col[:nv.value].copy(), idx[:ni.value].copy()


def _random_fills(seed):
    """Random concave fills: wobbly rings, stars, random (self-intersecting) polygons, holes and overlaps of either winding,
    both fill rules, fringes 0.5 .. 2."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(int(rs.randint(4, 10))):
        contours = []
        cx, cy = rs.uniform(100, 900), rs.uniform(100, 600)
        for _ in range(int(rs.randint(1, 4))):
            k = int(rs.randint(0, 3))
            ox, oy = cx + rs.uniform(-60, 60), cy + rs.uniform(-60, 60)
            if k == 0:
                contours.append(_ring(ox, oy, rs.uniform(15, 90), int(rs.randint(5, 40)), phase=rs.uniform(0, 6.28), cw=bool(rs.uniform() < 0.5),
                                      wobble=rs.uniform(0, 0.45), seed=int(rs.randint(0, 1 << 30))))
            elif k == 1:
                contours.append(_star(ox, oy, rs.uniform(40, 100), rs.uniform(10, 60), int(rs.randint(3, 9)), phase=rs.uniform(0, 6.28)))
            else:  # random polygon, usually self-intersecting
                n = int(rs.randint(3, 9))
                contours.append(np.stack([ox + rs.uniform(-90, 90, size=n), oy + rs.uniform(-90, 90, size=n)], axis=1).astype(np.float32))
        out.append((contours, int(rs.randint(0, 1 << 32, dtype=np.uint64)) | 0x10000000, float(rs.choice([0.5, 1.0, 1.5, 2.0])), int(rs.randint(0, 2))))
    return out


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_concave_fills_match_reference(rt, gpu_ctx, ref, seed):
    """Random polygons through strokerConcaveFillEndAA of the reference (libtess2 included, oracle/_ref) and through
    vgx_concave_move / vgx_concave_emit around the same libtess2 passes: bit-identical meshes."""
    _check_fills(rt, gpu_ctx, ref, _random_fills(100 + seed))


def test_concave_fringes_match_reference(rt, gpu_ctx, ref):
    _check_fills(rt, gpu_ctx, ref, _fills())


def _check_fills(rt, gpu_ctx, ref, fills):
    import torch
    capi = rt.capi
    # (1) caller: boundary contours of every fill
    tess, bverts, cont, frec = [], [], [], []
    vbase = 0
    for fi, (contours, color, fringe, eo) in enumerate(fills):
        t, verts, el = _tess_boundary(ref, contours, eo)
        tess.append((t, verts, el, eo))
        r = np.zeros(1, dtype=capi.concave_fill_dtype)
        r["first_contour"] = len(cont)
        r["num_contours"] = el.shape[0]
        r["color"] = color
        r["fringe"] = fringe
        frec.append(r)
        for first, n in el:
            c = np.zeros(1, dtype=capi.contour_dtype)
            c["first_vertex"] = vbase + int(first)
            c["num_vertices"] = int(n)
            c["fill"] = fi
            cont.append(c)
        # the contours of one fill must be back to back in tessGetElements order: that is how libtess2 returns them
        assert all(int(el[k, 0]) + int(el[k, 1]) == int(el[k + 1, 0]) for k in range(el.shape[0] - 1))
        bverts.append(verts)
        vbase += verts.shape[0]
    bverts = np.concatenate(bverts).astype(np.float32)
    cont = np.concatenate(cont)
    frec = np.concatenate(frec)
    dev = torch.device("cuda", 0)
    bv_d = torch.from_numpy(bverts).to(dev)
    cont_d = torch.from_numpy(cont.view(np.uint8).copy()).to(dev)
    # (2a) device: moved contours
    fr_d = torch.from_numpy(frec.view(np.uint8).copy()).to(dev)
    moved = rt.concave_move(gpu_ctx, bv_d, cont_d, cont.shape[0], fr_d, frec.shape[0]).cpu().numpy()
    # (3) caller: polygons of the moved contours
    tpos, tidx = [], []
    voff = 0
    for fi, (t, verts, el, eo) in enumerate(tess):
        pv, pi = _tess_polygons(ref, t, moved[voff:voff + verts.shape[0]], el, eo)
        frec["num_tess_vertices"][fi] = pv.shape[0]
        frec["num_tess_indices"][fi] = pi.shape[0]
        frec["first_tess_vertex"][fi] = sum(x.shape[0] for x in tpos)
        frec["first_tess_index"][fi] = sum(x.shape[0] for x in tidx)
        tpos.append(pv)
        tidx.append(pi)
        voff += verts.shape[0]
    tpos = np.concatenate(tpos).astype(np.float32)
    tidx = np.concatenate(tidx).astype(np.uint16)
    fr_d = torch.from_numpy(frec.view(np.uint8).copy()).to(dev)
    tp_d = torch.from_numpy(tpos if tpos.shape[0] else np.zeros((1, 2), np.float32)).to(dev)
    ti_d = torch.from_numpy(tidx.view(np.int16) if tidx.shape[0] else np.zeros(1, np.int16)).to(dev)
    # (2b) + (4) device: the meshes
    refs = [_reference_mesh(ref, c, col, fr, eo) for (c, col, fr, eo) in fills]
    nv = sum(r[0].shape[0] for r in refs)
    ni = sum(r[2].shape[0] for r in refs)
    bufs = rt.MeshBuffers(dev, nv, ni, len(fills))
    bufs.pos.fill_(float("nan"))
    rt.concave_emit(gpu_ctx, bv_d, cont_d, cont.shape[0], fr_d, frec.shape[0], tp_d, ti_d, bufs)
    torch.cuda.synchronize()
    assert int(bufs.dev_status.item()) == 0
    sz = bufs.dev_sizes.cpu().numpy()
    assert (int(sz[2]), int(sz[3]), int(sz[4])) == (len(fills), nv, ni)
    meshes = bufs.meshes[:len(fills) * 32].cpu().numpy().view(capi.mesh_dtype)
    pos = bufs.pos[:nv].cpu().numpy()
    col = bufs.color[:nv].cpu().numpy().view(np.uint32)
    idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
    v0 = i0 = 0
    for fi, (rp, rc, ri) in enumerate(refs):
        m = meshes[fi]
        assert (int(m["first_vertex"]), int(m["first_index"]), int(m["num_vertices"]), int(m["num_indices"])) == (v0, i0, rp.shape[0], ri.shape[0]), fi
        assert int(m["draw"]) == fi and int(m["subpath_kind"]) >> 28 == capi.MESH_CONCAVE_FILL_AA
        assert np.array_equal(idx[i0:i0 + ri.shape[0]], ri), ("idx", fi)
        assert np.array_equal(col[v0:v0 + rp.shape[0]], rc), ("color", fi)
        assert np.array_equal(pos[v0:v0 + rp.shape[0]].view(np.uint32), rp.view(np.uint32)), ("pos", fi, np.flatnonzero((pos[v0:v0 + rp.shape[0]] != rp).any(axis=1))[:5])
        v0 += rp.shape[0]
        i0 += ri.shape[0]
    # capacity is checked on the device
    small = rt.MeshBuffers(dev, nv // 2, ni, len(fills))
    rt.concave_emit(gpu_ctx, bv_d, cont_d, cont.shape[0], fr_d, frec.shape[0], tp_d, ti_d, small)
    torch.cuda.synchronize()
    assert int(small.dev_status.item()) == capi.VGX_E_NOSPACE
