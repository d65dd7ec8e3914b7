"""ctypes loader for the CPU oracles.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does (the HIP path fails loudly instead of falling back to it).

kind="reference": oracle/_ref/libvgref.so  = the reference's own src/path.cpp + src/stroker.cpp compiled
                  against oracle/bx_shim + csrc/vgmath.h (built by oracle/Makefile when /root/reference exists)
kind="port":      oracle/libvgoracle.so    = the restatement oracle/vgo_port.cpp
Parity status: pinned against the reference source itself run here; the bx transcendentals
(acos/atan2/cos/sin/tan/rsqrt) are UNPINNED upstream (bx not vendored, no version) and are defined by
csrc/vgmath.h for oracle and kernels alike.
"""
import ctypes as C
import importlib
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_pkg = importlib.import_module("vg-renderer_amd")
capi = _pkg.capi

_PATHS = {
    "reference": os.path.join(_HERE, "_ref", "libvgref.so"),
    "reference_sse": os.path.join(_HERE, "_ref", "libvgref_sse.so"),
    "reference_libm": os.path.join(_HERE, "_ref", "libvgref_libm.so"),  # sensitivity probe only (libm_sensitivity.py)
    "port": os.path.join(_HERE, "libvgoracle.so"),
}
_libs = {}


def available(kind):
    return os.path.exists(_PATHS[kind])


def default_kind():
    """The checker the tests use when they do not name one: the reference's own sources (oracle/_ref, built from
    /root/reference and shipped to the GPU box as a prebuilt .so) when present, else the restatement."""
    return "reference" if available("reference") else "port"


def load(kind=None):
    kind = kind or default_kind()
    if kind in _libs:
        return _libs[kind]
    lib = C.CDLL(_PATHS[kind])
    lib.vgo_flatten.restype = C.c_int
    lib.vgo_flatten.argtypes = [C.POINTER(capi.PathSetDesc), C.c_void_p, C.c_uint64, C.c_int, C.POINTER(capi.FlatOut), C.POINTER(capi.Sizes)]
    lib.vgo_tessellate.restype = C.c_int
    lib.vgo_tessellate.argtypes = [C.POINTER(capi.PathSetDesc), C.c_void_p, C.c_uint64, C.POINTER(capi.FlatOut), C.POINTER(capi.MeshOut), C.POINTER(capi.Sizes)]
    lib.vgo_assemble.restype = C.c_int
    lib.vgo_assemble.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
    lib.vgo_cache_localize.restype = C.c_int
    lib.vgo_cache_localize.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.vgo_cache_submit.restype = C.c_int
    lib.vgo_cache_submit.argtypes = [C.POINTER(capi.CacheDesc), C.c_void_p, C.c_uint64, C.POINTER(capi.MeshOut), C.POINTER(capi.Sizes)]
    lib.vgo_engine_name.restype = C.c_char_p
    _libs[kind] = lib
    return lib


class FlatResult:
    pass


class MeshResult:
    pass


def flatten(ps, draws, apply_transform=False, kind=None):
    """ps: PathSetArrays, draws: ndarray(draw_dtype). Returns FlatResult with numpy arrays."""
    lib = load(kind)
    draws = np.ascontiguousarray(draws)
    n = draws.shape[0]
    desc = ps.desc()
    sizes = capi.Sizes()
    st = lib.vgo_flatten(C.byref(desc), draws.ctypes.data, n, int(apply_transform), None, C.byref(sizes))
    assert st == 0, st
    r = FlatResult()
    r.poly = np.zeros((sizes.num_poly_vertices, 2), dtype=np.float32)
    r.subpaths = np.zeros(sizes.num_subpaths, dtype=capi.subpath_dtype)
    r.draw_info = np.zeros(n, dtype=capi.draw_info_dtype)
    fo = capi.FlatOut(r.poly.ctypes.data, r.subpaths.ctypes.data, r.draw_info.ctypes.data, sizes.num_poly_vertices, sizes.num_subpaths)
    st = lib.vgo_flatten(C.byref(desc), draws.ctypes.data, n, int(apply_transform), C.byref(fo), C.byref(sizes))
    assert st == 0, st
    r.sizes = sizes.as_dict()
    return r


def tessellate(ps, draws, kind=None, want_flat=False, count_only=False):
    lib = load(kind)
    draws = np.ascontiguousarray(draws)
    n = draws.shape[0]
    desc = ps.desc()
    sizes = capi.Sizes()
    st = lib.vgo_tessellate(C.byref(desc), draws.ctypes.data, n, None, None, C.byref(sizes))
    assert st == 0, st
    r = MeshResult()
    r.sizes = sizes.as_dict()
    if count_only:
        return r
    r.pos = np.zeros((sizes.num_vertices, 2), dtype=np.float32)
    r.color = np.zeros(sizes.num_vertices, dtype=np.uint32)
    r.idx = np.zeros(sizes.num_indices, dtype=np.uint16)
    r.meshes = np.zeros(sizes.num_meshes, dtype=capi.mesh_dtype)
    mo = capi.MeshOut(r.pos.ctypes.data, r.color.ctypes.data, r.idx.ctypes.data, r.meshes.ctypes.data,
                      sizes.num_vertices, sizes.num_indices, sizes.num_meshes)
    fo_ref = None
    if want_flat:
        r.poly = np.zeros((sizes.num_poly_vertices, 2), dtype=np.float32)
        r.subpaths = np.zeros(sizes.num_subpaths, dtype=capi.subpath_dtype)
        r.draw_info = np.zeros(n, dtype=capi.draw_info_dtype)
        fo = capi.FlatOut(r.poly.ctypes.data, r.subpaths.ctypes.data, r.draw_info.ctypes.data, sizes.num_poly_vertices, sizes.num_subpaths)
        fo_ref = C.byref(fo)
    st = lib.vgo_tessellate(C.byref(desc), draws.ctypes.data, n, fo_ref, C.byref(mo), C.byref(sizes))
    assert st == 0, st
    return r


def assemble(meshes, idx, max_vb_vertices=0, kind=None, mesh_keys=None):
    """Draw-command assembly (vgo_assemble): returns (status, drawcmds ndarray, rebased index buffer).
    mesh_keys: uint32 per mesh (its draw's state_key), None = one draw state."""
    lib = load(kind)
    meshes = np.ascontiguousarray(meshes)
    idx = np.ascontiguousarray(idx)
    out = np.zeros_like(idx)
    n = C.c_uint64(0)
    cap = max(int(meshes.shape[0]), 1)
    cmds = np.zeros(cap, dtype=capi.drawcmd_dtype)
    keys = None if mesh_keys is None else np.ascontiguousarray(mesh_keys, dtype=np.uint32)
    st = lib.vgo_assemble(meshes.ctypes.data, meshes.shape[0], idx.ctypes.data, out.ctypes.data, max_vb_vertices, cmds.ctypes.data, cap, C.byref(n),
                          None if keys is None else keys.ctypes.data)
    return st, cmds[:n.value], out


def cache_localize(draws, res, kind=None):
    """In place: res.pos <- local space (addCachedCommand). res: MeshResult of tessellate(ps, draws)."""
    lib = load(kind)
    draws = np.ascontiguousarray(draws)
    st = lib.vgo_cache_localize(draws.ctypes.data, draws.shape[0], res.pos.ctypes.data, res.meshes.ctypes.data, res.meshes.shape[0])
    assert st == 0, st
    return res


def cache_submit(res, instances, kind=None):
    """res: localised MeshResult; instances: ndarray(cache_instance_dtype). Returns a MeshResult of the frame."""
    lib = load(kind)
    instances = np.ascontiguousarray(instances)
    cd = capi.CacheDesc(res.pos.ctypes.data, res.color.ctypes.data, res.idx.ctypes.data, res.meshes.ctypes.data,
                        res.meshes.shape[0], res.pos.shape[0], res.idx.shape[0])
    sizes = capi.Sizes()
    st = lib.vgo_cache_submit(C.byref(cd), instances.ctypes.data, instances.shape[0], None, C.byref(sizes))
    assert st == 0, st
    r = MeshResult()
    r.sizes = sizes.as_dict()
    r.pos = np.zeros((sizes.num_vertices, 2), dtype=np.float32)
    r.color = np.zeros(sizes.num_vertices, dtype=np.uint32)
    r.idx = np.zeros(sizes.num_indices, dtype=np.uint16)
    r.meshes = np.zeros(sizes.num_meshes, dtype=capi.mesh_dtype)
    mo = capi.MeshOut(r.pos.ctypes.data, r.color.ctypes.data, r.idx.ctypes.data, r.meshes.ctypes.data, sizes.num_vertices, sizes.num_indices, sizes.num_meshes)
    st = lib.vgo_cache_submit(C.byref(cd), instances.ctypes.data, instances.shape[0], C.byref(mo), C.byref(sizes))
    assert st == 0, st
    return r


def tessellate_timed(ps, draws, kind=None, reps=1):
    """Time `reps` single-pass vgo_tessellate calls into preallocated (already touched) buffers.
    Returns (seconds, sizes dict). Used by bench.py's cpu_baseline leg."""
    import time
    lib = load(kind)
    draws = np.ascontiguousarray(draws)
    n = draws.shape[0]
    desc = ps.desc()
    sizes = capi.Sizes()
    st = lib.vgo_tessellate(C.byref(desc), draws.ctypes.data, n, None, None, C.byref(sizes))
    assert st == 0, st
    pos = np.ones((sizes.num_vertices, 2), dtype=np.float32)
    color = np.ones(sizes.num_vertices, dtype=np.uint32)
    idx = np.ones(sizes.num_indices, dtype=np.uint16)
    meshes = np.zeros(sizes.num_meshes, dtype=capi.mesh_dtype)
    mo = capi.MeshOut(pos.ctypes.data, color.ctypes.data, idx.ctypes.data, meshes.ctypes.data, sizes.num_vertices, sizes.num_indices, sizes.num_meshes)
    s2 = capi.Sizes()
    st = lib.vgo_tessellate(C.byref(desc), draws.ctypes.data, n, None, C.byref(mo), C.byref(s2))  # warm-up
    assert st == 0, st
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.vgo_tessellate(C.byref(desc), draws.ctypes.data, n, None, C.byref(mo), C.byref(s2))
    return time.perf_counter() - t0, sizes.as_dict()


def flatten_timed(ps, draws, kind=None, reps=1, apply_transform=True):
    """Time `reps` vgo_flatten calls (pathReset + commands + transformPath per draw) into preallocated buffers.
    Returns (seconds, sizes dict). Used by bench.py's cpu_baseline leg of the flatten-only config."""
    import time
    lib = load(kind)
    draws = np.ascontiguousarray(draws)
    n = draws.shape[0]
    desc = ps.desc()
    sizes = capi.Sizes()
    st = lib.vgo_flatten(C.byref(desc), draws.ctypes.data, n, int(apply_transform), None, C.byref(sizes))
    assert st == 0, st
    poly = np.ones((sizes.num_poly_vertices, 2), dtype=np.float32)
    subs = np.zeros(sizes.num_subpaths, dtype=capi.subpath_dtype)
    dinfo = np.zeros(n, dtype=capi.draw_info_dtype)
    fo = capi.FlatOut(poly.ctypes.data, subs.ctypes.data, dinfo.ctypes.data, sizes.num_poly_vertices, sizes.num_subpaths)
    s2 = capi.Sizes()
    st = lib.vgo_flatten(C.byref(desc), draws.ctypes.data, n, int(apply_transform), C.byref(fo), C.byref(s2))  # warm-up
    assert st == 0, st
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.vgo_flatten(C.byref(desc), draws.ctypes.data, n, int(apply_transform), C.byref(fo), C.byref(s2))
    return time.perf_counter() - t0, sizes.as_dict()
