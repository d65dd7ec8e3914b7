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


def run_async(rt, ctx, ps, d, shrink=None, profile=False):
    """vgx_tessellate_count (sizes + scratch) then the steady-state entry point vgx_tessellate into exactly sized
    buffers. Returns an object with sizes / status / pos / color / idx / meshes and `stages` (names of the kernels'
    profiling stages when profile=True)."""
    import torch
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    if shrink:
        bufs = rt.MeshBuffers(dd.device, int(nv * shrink), int(ni * shrink), nm)
    else:
        bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    bufs.pos.fill_(float("nan"))
    bufs.idx.fill_(-1)
    if profile:
        ctx.set_profiling(True)
    rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs)
    torch.cuda.synchronize()

    class G:
        pass
    got = G()
    got.stages = [n for n, _ in ctx.stage_times()] if profile else None
    if profile:
        ctx.set_profiling(False)
    got.status = int(bufs.dev_status.item())
    got.failure = ctx.failure_info() if got.status != 0 else None
    got.sizes = dict(sizes)
    dev = bufs.dev_sizes.cpu().numpy()
    got.dev_sizes = {k: int(dev[i]) for i, k in enumerate(
        ["num_poly_vertices", "num_subpaths", "num_meshes", "num_vertices", "num_indices", "num_serial_draws",
         "num_cmd_instances", "num_elements", "num_fill_elements", "num_drawcmds"])}
    if not shrink:
        got.pos = bufs.pos[:nv].cpu().numpy()
        got.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
        got.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
        got.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype)
    pset.close()
    return got


def describe_mesh_diff(got, ref):
    """Which mesh holds the first difference (for failure messages: a GPU run is expensive, say as much as possible)."""
    out = {}
    for k in ("first_vertex", "first_index", "num_vertices", "num_indices", "draw", "subpath_kind"):
        if got.meshes.shape == ref.meshes.shape and not np.array_equal(got.meshes[k], ref.meshes[k]):
            w = np.flatnonzero(got.meshes[k] != ref.meshes[k])
            out["meshes." + k] = (int(w[0]), int(w.shape[0]), int(got.meshes[k][w[0]]), int(ref.meshes[k][w[0]]))
    if got.meshes.shape != ref.meshes.shape:
        out["num_meshes"] = (got.meshes.shape[0], ref.meshes.shape[0])
        return out
    m = ref.meshes
    if got.pos.shape == ref.pos.shape:
        w = np.flatnonzero((got.pos.view(np.uint32) != ref.pos.view(np.uint32)).any(axis=1))
        if w.shape[0]:
            mi = int(np.searchsorted(m["first_vertex"], w[0], side="right") - 1)
            out["pos"] = {"first_vertex": int(w[0]), "count": int(w.shape[0]), "mesh": mi, "mesh_rec": {k: int(m[k][mi]) for k in m.dtype.names},
                          "got": got.pos[w[0]].tolist(), "ref": ref.pos[w[0]].tolist()}
    if got.idx.shape == ref.idx.shape:
        w = np.flatnonzero(got.idx != ref.idx)
        if w.shape[0]:
            mi = int(np.searchsorted(m["first_index"], w[0], side="right") - 1)
            out["idx"] = {"first_index": int(w[0]), "count": int(w.shape[0]), "mesh": mi, "mesh_rec": {k: int(m[k][mi]) for k in m.dtype.names},
                          "got": got.idx[w[0]:w[0] + 6].tolist(), "ref": ref.idx[w[0]:w[0] + 6].tolist()}
    if got.color.shape == ref.color.shape:
        w = np.flatnonzero(got.color != ref.color)
        if w.shape[0]:
            out["color"] = {"first": int(w[0]), "count": int(w.shape[0])}
    return out
