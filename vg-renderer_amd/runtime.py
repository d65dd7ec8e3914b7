"""Python plumbing over the C-ABI (libvgx.so): context, path sets, batch calls on torch device memory.

PyTorch is used only for device allocations and streams; every geometry computation happens in the
HIP kernels behind include/vgx.h. There is NO CPU fallback: if libvgx.so is missing or no gfx950 device is
present, construction raises.
"""
import ctypes as C
import os
import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VGX_LIB", os.path.join(_HERE, "libvgx.so"))  # VGX_LIB: tuning experiments only
_lib = None


class VgxError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        name = lib().vgx_status_string(status).decode() if _lib is not None else str(status)
        super().__init__("%s failed: %s (%d)" % (where, name, status))


def lib():
    """Load libvgx.so (built in-tree by `__graft_entry__.build()` / csrc/Makefile). Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libvgx.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` (HIP extension is mandatory, there is no CPU fallback)")
        # import torch first so that its HIP runtime (same SONAME) is the one both sides use
        import torch  # noqa: F401
        _lib = capi.bind(C.CDLL(LIB_PATH), capi.VGX_SYMBOLS)
    return _lib


def _check(st, where):
    if st != capi.VGX_OK:
        raise VgxError(st, where)


def validate_pathset(ps):
    """Host-only grammar / finiteness validation (no device needed)."""
    d = ps.desc()
    return lib().vgx_pathset_validate(C.byref(d))


class Context:
    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: vg-renderer_amd has no CPU fallback")
        self.device = device
        self._h = C.c_void_p()
        torch.cuda.set_device(device)
        torch.zeros(1, device="cuda:%d" % device)  # make sure the primary context exists
        _check(lib().vgx_create(device, C.byref(self._h)), "vgx_create")

    def close(self):
        if self._h:
            lib().vgx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def scratch_bytes(self):
        return int(lib().vgx_scratch_bytes(self._h))

    def set_profiling(self, on):
        _check(lib().vgx_set_profiling(self._h, 1 if on else 0), "vgx_set_profiling")

    def set_static_batches(self, on):
        """vgx_set_static_batches: batches keep their structure between counts (only transforms / colours move): the whole draw
        list becomes one template at the next tessellate_count, a step is then one kernel; a structural change -> VGX_E_STALE."""
        _check(lib().vgx_set_static_batches(self._h, 1 if on else 0), "vgx_set_static_batches")

    def set_assembly(self, drawcmds=None, max_vb_vertices=0, dev_num=None, split_state=False, uv=None, uv_value=None):
        """Arms draw-command assembly (vgx_set_assembly) with a uint8 device tensor of 48-byte vgx_drawcmd records,
        or disarms it (drawcmds=None). split_state: VGX_ASM_SPLIT_STATE. uv: device tensor of the UV stream (int16 [n,2]
        = 4 bytes per vertex, or float32 [n,2] = 8), uv_value: the white-pixel UV as a tuple of raw uint32 words.
        The tensors must stay alive while armed."""
        if drawcmds is None:
            _check(lib().vgx_set_assembly(self._h, None), "vgx_set_assembly")
            self._asm_keep = None
            return
        a = capi.Assembly()
        a.drawcmds = drawcmds.data_ptr()
        a.cap_drawcmds = drawcmds.numel() // capi.drawcmd_dtype.itemsize
        a.dev_num_drawcmds = dev_num.data_ptr() if dev_num is not None else None
        a.max_vb_vertices = max_vb_vertices
        a.flags = capi.ASM_SPLIT_STATE if split_state else 0
        a.reserved = 0
        if uv is not None:
            a.uv = uv.data_ptr()
            a.uv_bytes = uv.element_size() * 2
            a.uv_value[0] = int(uv_value[0]) & 0xFFFFFFFF
            a.uv_value[1] = int(uv_value[1]) & 0xFFFFFFFF if len(uv_value) > 1 else 0
        _check(lib().vgx_set_assembly(self._h, C.byref(a)), "vgx_set_assembly")
        self._asm_keep = (drawcmds, dev_num, uv)

    def failure_info(self):
        """Device status + why the single-pass kernel gave up, if it did (vgx_get_failure_info; synchronises)."""
        fi = capi.FailureInfo()
        _check(lib().vgx_get_failure_info(self._h, C.byref(fi), _stream_ptr()), "vgx_get_failure_info")
        return fi.as_dict()

    def stage_times(self, ncalls=1):
        """Per-kernel HIP-event times of the last profiled call, or their average over the last `ncalls` calls."""
        st = capi.StageTimes()
        _check(lib().vgx_get_stage_times_avg(self._h, C.byref(st), int(ncalls)), "vgx_get_stage_times_avg")
        return [(st.name[i].decode(), float(st.ms[i])) for i in range(st.num_stages)]


class PathSet:
    """Path definitions resident in HBM (vgx_pathset)."""

    def __init__(self, ctx, arrays):
        self.ctx = ctx
        self.arrays = arrays
        self._h = C.c_void_p()
        d = arrays.desc()
        _check(lib().vgx_pathset_create(ctx.handle, C.byref(d), C.byref(self._h)), "vgx_pathset_create")

    def close(self):
        if self._h and self.ctx.handle:
            lib().vgx_pathset_destroy(self.ctx.handle, self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h


def pin_draws(draws):
    """numpy draw records -> uint8 torch tensor in PINNED host memory: where a caller that cares about the upload writes its draw
    records in the first place (pageable memory reaches the device at 7-20 GB/s through the runtime's staging, pinned at ~55)."""
    import torch
    raw = np.ascontiguousarray(draws).view(np.uint8).reshape(-1)
    t = torch.empty(raw.shape[0], dtype=torch.uint8, pin_memory=True)
    t.numpy()[:] = raw
    return t


def upload_draws(draws, device=0):
    """draw records (numpy, or a pinned uint8 tensor from pin_draws) -> uint8 torch tensor in HBM (64 bytes per draw)."""
    import torch
    if isinstance(draws, torch.Tensor):
        return draws.to("cuda:%d" % device, non_blocking=True)
    raw = np.ascontiguousarray(draws).view(np.uint8).reshape(-1)
    return torch.from_numpy(raw).to("cuda:%d" % device)


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class FlatResult:
    pass


class MeshResult:
    pass


def flatten(ctx, pset, draws_dev, ndraws, apply_transform=False, to_host=True, entry=None):
    """The flatten entry points. draws_dev: uint8 torch tensor from upload_draws.
    entry "two_phase": vgx_flatten_count + vgx_flatten_emit; "one_walk": vgx_flatten into buffers of exactly the counted sizes;
    "both" (default; VGX_FLATTEN_ENTRY overrides): the two-phase result, after checking that vgx_flatten produced the same
    bytes -- polyline, sub-path records, per-draw records, totals -- so that every caller of this helper pins both entry points."""
    import torch
    entry = entry or os.environ.get("VGX_FLATTEN_ENTRY", "both")
    if entry in ("one_walk", "both"):
        r2 = _flatten_two_phase(ctx, pset, draws_dev, ndraws, apply_transform, to_host=False)
        r1 = flatten_one_walk(ctx, pset, draws_dev, ndraws, apply_transform, cap_poly=r2.sizes["num_poly_vertices"], cap_subs=r2.sizes["num_subpaths"], to_host=to_host)
        if entry == "both":
            for k in ("num_poly_vertices", "num_subpaths", "num_meshes", "num_serial_draws", "num_cmd_instances"):
                assert r1.sizes[k] == r2.sizes[k], ("vgx_flatten vs two-phase", k, r1.sizes[k], r2.sizes[k])
            npv, nsp = r2.sizes["num_poly_vertices"], r2.sizes["num_subpaths"]
            assert torch.equal(r1.poly_dev[:npv].view(torch.int32), r2.poly_dev[:npv].view(torch.int32)), "vgx_flatten: polyline differs from the two-phase entry"
            assert torch.equal(r1.subs_dev[:nsp * 16], r2.subs_dev[:nsp * 16]), "vgx_flatten: sub-path records differ from the two-phase entry"
            assert torch.equal(r1.dinfo_dev[:ndraws * 40], r2.dinfo_dev[:ndraws * 40]), "vgx_flatten: per-draw records differ from the two-phase entry"
        return r1
    return _flatten_two_phase(ctx, pset, draws_dev, ndraws, apply_transform, to_host)


def _flatten_two_phase(ctx, pset, draws_dev, ndraws, apply_transform=False, to_host=True):
    import torch
    L = lib()
    sizes = capi.Sizes()
    s = _stream_ptr()
    _check(L.vgx_flatten_count(ctx.handle, pset.handle, draws_dev.data_ptr(), ndraws, C.byref(sizes), s), "vgx_flatten_count")
    dev = draws_dev.device
    npv, nsp = int(sizes.num_poly_vertices), int(sizes.num_subpaths)
    poly = torch.empty((max(npv, 1), 2), dtype=torch.float32, device=dev)
    subs = torch.empty(max(nsp, 1) * 16, dtype=torch.uint8, device=dev)
    dinfo = torch.empty(max(ndraws, 1) * 40, dtype=torch.uint8, device=dev)
    out = capi.FlatOut(poly.data_ptr(), subs.data_ptr(), dinfo.data_ptr(), npv, nsp)
    _check(L.vgx_flatten_emit(ctx.handle, pset.handle, draws_dev.data_ptr(), ndraws, int(apply_transform), C.byref(out), s), "vgx_flatten_emit")
    torch.cuda.synchronize()
    r = FlatResult()
    r.sizes = sizes.as_dict()
    r.poly_dev, r.subs_dev, r.dinfo_dev = poly, subs, dinfo
    if to_host:
        r.poly = poly[:npv].cpu().numpy()
        r.subpaths = subs[:nsp * 16].cpu().numpy().view(capi.subpath_dtype)
        r.draw_info = dinfo[:ndraws * 40].cpu().numpy().view(capi.draw_info_dtype)
    return r


class FlatBuffers:
    """Caller-owned output buffers of vgx_flatten in HBM (vgx_flat_out) + the device-side totals / status words."""

    def __init__(self, device, npoly, nsubs, ndraws):
        import torch
        self.cap = (int(npoly), int(nsubs))
        self.poly = torch.empty((max(int(npoly), 1), 2), dtype=torch.float32, device=device)
        self.subs = torch.empty(max(int(nsubs), 1) * 16, dtype=torch.uint8, device=device)
        self.dinfo = torch.empty(max(int(ndraws), 1) * 40, dtype=torch.uint8, device=device)
        self.dev_sizes = torch.zeros(10, dtype=torch.int64, device=device)
        self.dev_status = torch.zeros(1, dtype=torch.int32, device=device)

    def out_struct(self):
        return capi.FlatOut(self.poly.data_ptr(), self.subs.data_ptr(), self.dinfo.data_ptr(), self.cap[0], self.cap[1])


def flatten_async(ctx, pset, draws_dev, ndraws, bufs, apply_transform=False):
    """vgx_flatten: the ordered one-walk flatten, single asynchronous call; totals / status land in bufs.dev_*."""
    out = bufs.out_struct()
    _check(lib().vgx_flatten(ctx.handle, pset.handle, draws_dev.data_ptr(), ndraws, int(apply_transform), C.byref(out),
                             bufs.dev_sizes.data_ptr(), bufs.dev_status.data_ptr(), _stream_ptr()), "vgx_flatten")


def flatten_one_walk(ctx, pset, draws_dev, ndraws, apply_transform=False, cap_poly=None, cap_subs=None, to_host=True):
    """vgx_flatten into buffers of the given capacities (default: generous), results like `flatten`. Raises VgxError with the
    device status when it is not VGX_OK (the sizes of the failed call stay available as `.sizes` on the exception)."""
    import torch
    dev = draws_dev.device
    if cap_poly is None or cap_subs is None:
        # sizes from the two-phase entry: callers that know their capacities pass them
        z = capi.Sizes()
        _check(lib().vgx_flatten_count(ctx.handle, pset.handle, draws_dev.data_ptr(), ndraws, C.byref(z), _stream_ptr()), "vgx_flatten_count")
        cap_poly = int(z.num_poly_vertices) if cap_poly is None else cap_poly
        cap_subs = int(z.num_subpaths) if cap_subs is None else cap_subs
    bufs = FlatBuffers(dev, cap_poly, cap_subs, ndraws)
    flatten_async(ctx, pset, draws_dev, ndraws, bufs, apply_transform)
    torch.cuda.synchronize()
    st = int(bufs.dev_status.item())
    z = bufs.dev_sizes.cpu().numpy()
    names = [k for k, _ in capi.Sizes._fields_]
    sizes = {k: int(z[i]) for i, k in enumerate(names)}
    if st != capi.VGX_OK:
        e = VgxError(st, "vgx_flatten")
        e.sizes = sizes
        raise e
    r = FlatResult()
    r.sizes = sizes
    r.poly_dev, r.subs_dev, r.dinfo_dev = bufs.poly, bufs.subs, bufs.dinfo
    if to_host:
        npv, nsp = sizes["num_poly_vertices"], sizes["num_subpaths"]
        r.poly = bufs.poly[:npv].cpu().numpy()
        r.subpaths = bufs.subs[:nsp * 16].cpu().numpy().view(capi.subpath_dtype)
        r.draw_info = bufs.dinfo[:ndraws * 40].cpu().numpy().view(capi.draw_info_dtype)
    return r


class MeshBuffers:
    """Caller-owned output buffers in HBM (vgx_mesh_out)."""

    def __init__(self, device, nverts, nidx, nmeshes):
        import torch
        self.cap = (int(nverts), int(nidx), int(nmeshes))
        self.pos = torch.empty((max(nverts, 1), 2), dtype=torch.float32, device=device)
        self.color = torch.empty(max(nverts, 1), dtype=torch.int32, device=device)
        self.idx = torch.empty(max(nidx, 1), dtype=torch.int16, device=device)
        self.meshes = torch.empty(max(nmeshes, 1) * 32, dtype=torch.uint8, device=device)
        self.dev_sizes = torch.zeros(10, dtype=torch.int64, device=device)
        self.dev_status = torch.zeros(1, dtype=torch.int32, device=device)

    def out_struct(self):
        return capi.MeshOut(self.pos.data_ptr(), self.color.data_ptr(), self.idx.data_ptr(), self.meshes.data_ptr(),
                            self.cap[0], self.cap[1], self.cap[2])

    def view(self, v0, nv, i0, ni, m0, nm):
        """A tile of these buffers: vertices [v0, v0 + nv), indices [i0, i0 + ni), mesh records [m0, m0 + nm) as buffers of
        their own (same memory; own totals / status words) -- what one sub-batch of a frame is tessellated into."""
        import torch
        t = MeshBuffers.__new__(MeshBuffers)
        t.cap = (int(nv), int(ni), int(nm))
        t.pos = self.pos[v0:v0 + max(nv, 1)]
        t.color = self.color[v0:v0 + max(nv, 1)]
        t.idx = self.idx[i0:i0 + max(ni, 1)]
        t.meshes = self.meshes[m0 * 32:(m0 + max(nm, 1)) * 32]
        t.dev_sizes = torch.zeros(10, dtype=torch.int64, device=self.pos.device)
        t.dev_status = torch.zeros(1, dtype=torch.int32, device=self.pos.device)
        return t


def partition(ctx, pset, draws_dev, ndraws, nparts):
    """vgx_partition: contiguous draw ranges of about equal predicted output. Returns (bounds [nparts + 1], weights [nparts])."""
    bounds = (C.c_uint64 * (nparts + 1))()
    weights = (C.c_uint64 * nparts)()
    _check(lib().vgx_partition(ctx.handle, pset.handle, draws_dev.data_ptr(), ndraws, nparts, bounds, weights, _stream_ptr()), "vgx_partition")
    return list(bounds), list(weights)


def tessellate_count(ctx, pset, draws_dev, ndraws):
    sizes = capi.Sizes()
    _check(lib().vgx_tessellate_count(ctx.handle, pset.handle, draws_dev.data_ptr(), ndraws, C.byref(sizes), _stream_ptr()), "vgx_tessellate_count")
    return sizes.as_dict()


def tessellate_emit(ctx, pset, draws_dev, ndraws, bufs):
    out = bufs.out_struct()
    _check(lib().vgx_tessellate_emit(ctx.handle, pset.handle, draws_dev.data_ptr(), ndraws, C.byref(out), _stream_ptr()), "vgx_tessellate_emit")


def tessellate_async(ctx, pset, draws_dev, ndraws, bufs):
    """Steady-state call: whole pipeline, no host round trip; totals/status land in bufs.dev_*."""
    out = bufs.out_struct()
    _check(lib().vgx_tessellate(ctx.handle, pset.handle, draws_dev.data_ptr(), ndraws, C.byref(out),
                                bufs.dev_sizes.data_ptr(), bufs.dev_status.data_ptr(), _stream_ptr()), "vgx_tessellate")


def tessellate(ctx, pset, draws_dev, ndraws, to_host=True):
    """count -> allocate exact -> emit. Returns MeshResult (numpy copies when to_host)."""
    import torch
    sizes = tessellate_count(ctx, pset, draws_dev, ndraws)
    bufs = MeshBuffers(draws_dev.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    tessellate_emit(ctx, pset, draws_dev, ndraws, bufs)
    torch.cuda.synchronize()
    r = MeshResult()
    r.sizes = sizes
    r.bufs = bufs
    if to_host:
        nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
        r.pos = bufs.pos[:nv].cpu().numpy()
        r.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
        r.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
        r.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(capi.mesh_dtype)
    return r


def stroke(ctx, poly_dev, subs_dev, subdraw_dev, nsubs, draws_dev, ndraws, to_host=True):
    """Stroker-level entry (vgx_stroke_count + vgx_stroke_emit): already flattened + transformed vertex lists in,
    meshes out. poly_dev float32 [n,2], subs_dev uint8 (16-byte vgx_subpath records), subdraw_dev int32 [nsubs]."""
    import torch
    L = lib()
    sizes = capi.Sizes()
    s = _stream_ptr()
    _check(L.vgx_stroke_count(ctx.handle, poly_dev.data_ptr(), subs_dev.data_ptr(), subdraw_dev.data_ptr(), nsubs, draws_dev.data_ptr(), ndraws, C.byref(sizes), s), "vgx_stroke_count")
    sz = sizes.as_dict()
    bufs = MeshBuffers(poly_dev.device, sz["num_vertices"], sz["num_indices"], sz["num_meshes"])
    out = bufs.out_struct()
    _check(L.vgx_stroke_emit(ctx.handle, poly_dev.data_ptr(), subs_dev.data_ptr(), subdraw_dev.data_ptr(), nsubs, draws_dev.data_ptr(), ndraws, C.byref(out), s), "vgx_stroke_emit")
    torch.cuda.synchronize()
    r = MeshResult()
    r.sizes = sz
    r.bufs = bufs
    if to_host:
        nv, ni, nm = sz["num_vertices"], sz["num_indices"], sz["num_meshes"]
        r.pos = bufs.pos[:nv].cpu().numpy()
        r.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
        r.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16)
        r.meshes = bufs.meshes[:nm * 32].cpu().numpy().view(capi.mesh_dtype)
    return r


# ---- shape cache (vgx_cache_localize / vgx_cache_submit) ---------------------------------------------
class MeshCache:
    """A tessellated drawing kept in HBM in local space (the reference's CommandListCache, vg.cpp:249-256)."""

    def __init__(self, ctx, bufs, sizes, draws_dev, ndraws):
        """bufs: MeshBuffers that vgx_tessellate[_emit] filled for `draws_dev`; positions are localised in place."""
        self.bufs = bufs
        self.nv, self.ni, self.nm = int(sizes["num_vertices"]), int(sizes["num_indices"]), int(sizes["num_meshes"])
        _check(lib().vgx_cache_localize(ctx.handle, draws_dev.data_ptr(), ndraws, bufs.pos.data_ptr(), bufs.meshes.data_ptr(), self.nm, _stream_ptr()), "vgx_cache_localize")

    def desc(self):
        b = self.bufs
        return capi.CacheDesc(b.pos.data_ptr(), b.color.data_ptr(), b.idx.data_ptr(), b.meshes.data_ptr(), self.nm, self.nv, self.ni)


def cache_submit(ctx, cache, instances_dev, ninst, bufs):
    """instances_dev: uint8 device tensor of 40-byte vgx_cache_instance records. Asynchronous."""
    d = cache.desc()
    out = bufs.out_struct()
    _check(lib().vgx_cache_submit(ctx.handle, C.byref(d), instances_dev.data_ptr(), ninst, C.byref(out),
                                  bufs.dev_sizes.data_ptr(), bufs.dev_status.data_ptr(), _stream_ptr()), "vgx_cache_submit")


# ---- concave fills (vgx_concave_move / vgx_concave_emit): libtess2 stays with the caller -----------------------------
def concave_move(ctx, contour_verts_dev, contours_dev, ncontours, fills_dev, nfills):
    """Inner fringe vertex of every boundary-contour vertex (what the reference writes back into the contour before the
    second libtess2 pass). contour_verts_dev float32 [n,2]; contours_dev / fills_dev uint8 tensors of 16-byte vgx_contour /
    48-byte vgx_concave_fill records. Returns a float32 [n,2] device tensor."""
    import torch
    n = int(contour_verts_dev.shape[0])
    moved = torch.empty_like(contour_verts_dev)
    _check(lib().vgx_concave_move(ctx.handle, contour_verts_dev.data_ptr(), n, contours_dev.data_ptr(), ncontours,
                                  fills_dev.data_ptr(), nfills, moved.data_ptr(), _stream_ptr()), "vgx_concave_move")
    return moved


def concave_emit(ctx, contour_verts_dev, contours_dev, ncontours, fills_dev, nfills, tess_pos_dev, tess_idx_dev, bufs):
    """One mesh per fill: fringe + interior (see include/vgx.h). Asynchronous; totals / status land in bufs.dev_*."""
    n = int(contour_verts_dev.shape[0])
    out = bufs.out_struct()
    _check(lib().vgx_concave_emit(ctx.handle, contour_verts_dev.data_ptr(), n, contours_dev.data_ptr(), ncontours,
                                  fills_dev.data_ptr(), nfills, tess_pos_dev.data_ptr(), tess_idx_dev.data_ptr(), C.byref(out),
                                  bufs.dev_sizes.data_ptr(), bufs.dev_status.data_ptr(), _stream_ptr()), "vgx_concave_emit")


# ---- merging external meshes into a frame (vgx_merge) ---------------------------------------------------------------
def mesh_seq(bufs, nv, ni, nm):
    """A finished mesh sequence (what vgx_tessellate / vgx_concave_emit wrote into `bufs`) as a vgx_cache_desc."""
    return capi.CacheDesc(bufs.pos.data_ptr(), bufs.color.data_ptr(), bufs.idx.data_ptr(), bufs.meshes.data_ptr(), int(nm), int(nv), int(ni))


def merge(ctx, seq_a, seq_b, b_draw_dev, draws_dev, ndraws, bufs, b_uv_dev=None):
    """Both sequences interleaved by draw index into `bufs` (honours an armed assembly). b_draw_dev: int32 device tensor, the
    frame draw of every mesh of seq_b (or None). b_uv_dev: per-vertex UVs of seq_b (IndexedTriList meshes), copied into the
    armed assembly's UV stream. Asynchronous; totals / status land in bufs.dev_*."""
    out = bufs.out_struct()
    _check(lib().vgx_merge_uv(ctx.handle, C.byref(seq_a), C.byref(seq_b), b_draw_dev.data_ptr() if b_draw_dev is not None else None,
                              b_uv_dev.data_ptr() if b_uv_dev is not None else None,
                              draws_dev.data_ptr() if draws_dev is not None else None, ndraws, C.byref(out),
                              bufs.dev_sizes.data_ptr(), bufs.dev_status.data_ptr(), _stream_ptr()), "vgx_merge_uv")
