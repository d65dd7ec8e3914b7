"""Comparison helpers for the parity tests."""
import numpy as np


def bytes_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8).reshape(-1), b.view(np.uint8).reshape(-1))


def first_diff(a, b, n=5):
    a = np.ascontiguousarray(a).reshape(-1)
    b = np.ascontiguousarray(b).reshape(-1)
    m = min(a.shape[0], b.shape[0])
    w = np.flatnonzero(a[:m].view(np.uint8 if a.dtype.fields else a.dtype) != b[:m].view(np.uint8 if b.dtype.fields else b.dtype))[:n]
    return {"where": w.tolist(), "len": (a.shape[0], b.shape[0])}


def assert_flat_equal(got, ref, what=""):
    """Flatten outputs: polyline bit-exact, sub-path table and draw info identical."""
    assert got.sizes["num_poly_vertices"] == ref.sizes["num_poly_vertices"], (what, got.sizes, ref.sizes)
    assert got.sizes["num_subpaths"] == ref.sizes["num_subpaths"], (what, got.sizes, ref.sizes)
    for k in ("first_vertex", "num_vertices", "flags"):
        assert np.array_equal(got.subpaths[k], ref.subpaths[k]), (what, "subpaths." + k, first_diff(got.subpaths[k], ref.subpaths[k]))
    for k in ("first_poly_vertex", "first_subpath", "num_poly_vertices", "num_subpaths"):
        assert np.array_equal(got.draw_info[k], ref.draw_info[k]), (what, "draw_info." + k, first_diff(got.draw_info[k], ref.draw_info[k]))
    assert bytes_equal(got.poly, ref.poly), (what, "poly", first_diff(got.poly.view(np.uint32), ref.poly.view(np.uint32)))


def assert_mesh_equal(got, ref, what="", pos_tol=0.0):
    """Tessellation outputs: indices, colours and the mesh table bit-exact; positions bit-exact by
    default (pos_tol=0) -- the north-star bound is 1e-4, we hold the kernels to 0 ulp."""
    for k in ("num_meshes", "num_vertices", "num_indices"):
        assert got.sizes[k] == ref.sizes[k], (what, k, got.sizes, ref.sizes)
    for k in ("first_vertex", "first_index", "num_vertices", "num_indices", "draw", "subpath_kind"):
        assert np.array_equal(got.meshes[k], ref.meshes[k]), (what, "meshes." + k, first_diff(got.meshes[k], ref.meshes[k]))
    assert np.array_equal(got.idx, ref.idx), (what, "idx", first_diff(got.idx, ref.idx))
    assert np.array_equal(got.color, ref.color), (what, "color", first_diff(got.color, ref.color))
    if pos_tol == 0.0:
        assert bytes_equal(got.pos, ref.pos), (what, "pos", first_diff(got.pos.view(np.uint32), ref.pos.view(np.uint32)))
    else:
        assert np.max(np.abs(got.pos - ref.pos)) <= pos_tol, (what, "pos")
