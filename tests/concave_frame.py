"""The CALLER's side of a frame that contains concave fills (TEST INFRASTRUCTURE): libtess2 stays on the CPU -- here the
libtess2 that is linked into oracle/_ref/libvgref.so, driven the way strokerConcaveFillEnd[AA] drives it (reference
src/stroker.cpp:849-1006) -- and the product does everything else:
  vgx_cmdlist_decode      concave FillPath* commands -> draws without a GPU mesh (VGX_FILL_CONCAVE)
  vgx_tessellate          sequence A: the meshes of every other draw
  vgx_flatten_*           the transformed contours of the concave draws (pathXXX + transformPath)
  libtess2 (caller)       boundary contours / polygons
  vgx_concave_move/_emit  sequence B: fringe + interior of every concave fill
  vgx_merge               A and B interleaved by draw index, assembled into draw commands like any frame."""
import ctypes as C

import numpy as np

import test_gpu_concave as TC


def _polygons(ref, contours, even_odd):
    """strokerConcaveFillEnd (stroker.cpp:849-866): tessTesselate(TESS_POLYGONS) of the contours as they are."""
    t = ref.vgo_tess_new()
    for c in contours:
        c = np.ascontiguousarray(c, dtype=np.float32)
        ref.vgo_tess_add_contour(t, c.ctypes.data, c.shape[0])
    ref.vgo_tess_run_plain.restype = C.c_int
    ref.vgo_tess_run_plain.argtypes = [C.c_void_p, C.c_int]
    assert ref.vgo_tess_run_plain(t, even_odd) == 1  # no normal: the non-AA call of the reference
    nv, ne = ref.vgo_tess_vertex_count(t), ref.vgo_tess_element_count(t)
    verts = np.ctypeslib.as_array(ref.vgo_tess_vertices(t), shape=(nv, 2)).copy() if nv else np.zeros((0, 2), np.float32)
    idx = np.ctypeslib.as_array(ref.vgo_tess_elements(t), shape=(ne * 3,)).copy() if ne else np.zeros((0,), np.uint16)
    ref.vgo_tess_delete(t)
    return verts, idx


def gpu_frame(rt, ctx, ref, ps, draws, max_vb, uv_bytes=4, uv_value=0, tri=None):
    """Whole frame on the device + libtess2 on the host. Returns the dict tests/test_cmdlist_ref.py::gpu_frame returns."""
    import torch
    capi = rt.capi
    dev = torch.device("cuda", 0)
    n = draws.shape[0]
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(draws)
    # ---- sequence A: every draw that is not a concave fill (those have no VGX_FILL_ENABLE: no mesh)
    sa = rt.tessellate_count(ctx, pset, dd, n)
    A = rt.MeshBuffers(dev, sa["num_vertices"], sa["num_indices"], sa["num_meshes"])
    rt.tessellate_emit(ctx, pset, dd, n, A)
    # ---- the concave draws' contours: pathXXX + transformPath on the device
    cidx = np.flatnonzero((draws["fill_flags"] & capi.FILL_CONCAVE) != 0)
    fills, bverts, cont, frec, b_draw = [], [], [], [], []
    if cidx.shape[0]:
        dsub = rt.upload_draws(draws[cidx])
        fl = rt.flatten(ctx, pset, dsub, cidx.shape[0], apply_transform=True)
        vbase = 0
        for k, di in enumerate(cidx):
            info = fl.draw_info[k]
            subs = fl.subpaths[int(info["first_subpath"]):int(info["first_subpath"]) + int(info["num_subpaths"])]
            if subs.shape[0] == 0 or (subs["num_vertices"] < 3).any():
                continue  # `if (subPath->m_NumVertices < 3) return;` (vg.cpp:3139-3141): no mesh for this fill at all
            contours = [fl.poly[int(s["first_vertex"]):int(s["first_vertex"]) + int(s["num_vertices"])] for s in subs]
            ff = int(draws["fill_flags"][di])
            aa, eo = (ff & capi.FILL_AA) != 0, 1 if (ff & capi.FILL_EVEN_ODD) else 0
            r = np.zeros(1, dtype=capi.concave_fill_dtype)
            r["first_contour"] = len(cont)
            r["color"] = draws["fill_color"][di]
            r["fringe"] = draws["fringe"][di]
            if aa:  # strokerConcaveFillEndAA: boundary contours first
                t, verts, el = TC._tess_boundary(ref, contours, eo)
                r["num_contours"] = el.shape[0]
                for first, cnt in el:
                    c = np.zeros(1, dtype=capi.contour_dtype)
                    c["first_vertex"] = vbase + int(first)
                    c["num_vertices"] = int(cnt)
                    c["fill"] = len(frec)
                    cont.append(c)
                fills.append(("aa", t, verts, el, eo, vbase))
                bverts.append(verts)
                vbase += verts.shape[0]
            else:   # strokerConcaveFillEnd: polygons of the path's own contours, one colour
                fills.append(("plain", contours, eo))
            frec.append(r)
            b_draw.append(int(di))
    nf = len(frec)
    if nf:
        frec = np.concatenate(frec)
        bverts_np = np.concatenate(bverts).astype(np.float32) if bverts else np.zeros((1, 2), np.float32)
        cont_np = np.concatenate(cont) if cont else np.zeros(1, dtype=capi.contour_dtype)
        ncont = len(cont)
        bv_d = torch.from_numpy(bverts_np).to(dev)
        cont_d = torch.from_numpy(cont_np.view(np.uint8).copy()).to(dev)
        fr_d = torch.from_numpy(frec.view(np.uint8).copy()).to(dev)
        moved = rt.concave_move(ctx, bv_d, cont_d, ncont, fr_d, nf).cpu().numpy() if ncont else bverts_np
        tpos, tidx = [], []
        for fi, f in enumerate(fills):
            if f[0] == "aa":
                _, t, verts, el, eo, vb = f
                pv, pi = TC._tess_polygons(ref, t, moved[vb:vb + verts.shape[0]], el, eo)
            else:
                pv, pi = _polygons(ref, f[1], f[2])
            frec["num_tess_vertices"][fi] = pv.shape[0]
            frec["num_tess_indices"][fi] = pi.shape[0]
            frec["first_tess_vertex"][fi] = sum(x.shape[0] for x in tpos)
            frec["first_tess_index"][fi] = sum(x.shape[0] for x in tidx)
            tpos.append(pv)
            tidx.append(pi)
        tpos = np.concatenate(tpos).astype(np.float32)
        tidx = np.concatenate(tidx).astype(np.uint16)
        fr_d = torch.from_numpy(frec.view(np.uint8).copy()).to(dev)
        tp_d = torch.from_numpy(tpos if tpos.shape[0] else np.zeros((1, 2), np.float32)).to(dev)
        ti_d = torch.from_numpy(tidx.view(np.int16) if tidx.shape[0] else np.zeros(1, np.int16)).to(dev)
        # sizes of sequence B are known on the host: 2 / 6 per boundary-contour vertex + the interior
        cv = np.zeros(nf, np.int64)
        for c in cont:
            cv[int(c["fill"][0])] += int(c["num_vertices"][0])
        bnv = int((2 * cv + frec["num_tess_vertices"]).sum())
        bni = int((6 * cv + frec["num_tess_indices"]).sum())
        B = rt.MeshBuffers(dev, bnv, bni, nf)
        rt.concave_emit(ctx, bv_d, cont_d, ncont, fr_d, nf, tp_d, ti_d, B)
        torch.cuda.synchronize()
        assert int(B.dev_status.item()) == 0
        seq_b = rt.mesh_seq(B, bnv, bni, nf)
        bd = torch.from_numpy(np.asarray(b_draw, np.int32)).to(dev)
    else:
        B = rt.MeshBuffers(dev, 1, 1, 1)
        bnv = bni = 0
        seq_b = rt.mesh_seq(B, 0, 0, 0)
        bd = None
    seq_a = rt.mesh_seq(A, sa["num_vertices"], sa["num_indices"], sa["num_meshes"])
    anv, ani, anm = sa["num_vertices"], sa["num_indices"], sa["num_meshes"]
    b_uv = None
    if tri is not None and tri["meshes"].shape[0]:
        # ---- user meshes (IndexedTriList): the decoder's tri_* arrays, uploaded, are a third sequence. Both external
        # sequences are sorted by draw on their own, so: (A + concave) first, without assembly, then (that + user meshes)
        if nf:
            AB = rt.MeshBuffers(dev, anv + bnv, ani + bni, anm + nf)
            rt.merge(ctx, seq_a, seq_b, bd, dd, n, AB)
            torch.cuda.synchronize()
            assert int(AB.dev_status.item()) == 0
            anv, ani, anm = anv + bnv, ani + bni, anm + nf
            seq_a = rt.mesh_seq(AB, anv, ani, anm)
        bnv, bni, nfb = tri["pos"].shape[0], tri["idx"].shape[0], tri["meshes"].shape[0]
        B = rt.MeshBuffers(dev, bnv, bni, nfb)
        if bnv:
            B.pos[:bnv] = torch.from_numpy(tri["pos"]).to(dev)
            B.color[:bnv] = torch.from_numpy(tri["color"].view(np.int32)).to(dev)
            b_uv = torch.from_numpy(np.ascontiguousarray(tri["uv"])).to(dev)
        if bni:
            B.idx[:bni] = torch.from_numpy(tri["idx"].view(np.int16)).to(dev)
        B.meshes[:nfb * 32] = torch.from_numpy(tri["meshes"].view(np.uint8).copy()).to(dev)
        seq_b = rt.mesh_seq(B, bnv, bni, nfb)
        bd = None  # the records' own draw fields
    else:
        nfb = nf
    # ---- the frame: the sequences interleaved by draw, assembled
    nv, ni, nm = anv + bnv, ani + bni, anm + nfb
    out = rt.MeshBuffers(dev, nv, ni, nm)
    cmds = torch.zeros((nm + 2) * 48, dtype=torch.uint8, device=dev)
    ncmd = torch.zeros(1, dtype=torch.int64, device=dev)
    uv = torch.zeros((max(nv, 1), 2), dtype=torch.int16 if uv_bytes == 4 else torch.float32, device=dev)
    uvw = uv_value if isinstance(uv_value, (tuple, list)) else (uv_value, 0)
    ctx.set_assembly(cmds, max_vb, ncmd, split_state=True, uv=uv, uv_value=tuple(int(x) for x in uvw))
    try:
        rt.merge(ctx, seq_a, seq_b, bd, dd, n, out, b_uv_dev=b_uv)
        torch.cuda.synchronize()
    finally:
        ctx.set_assembly(None)
    assert int(out.dev_status.item()) == 0, int(out.dev_status.item())
    sz = out.dev_sizes.cpu().numpy()
    assert (int(sz[2]), int(sz[3]), int(sz[4])) == (nm, nv, ni)
    k = int(ncmd.item())
    res = dict(cmds=cmds[:k * 48].cpu().numpy().view(capi.drawcmd_dtype), idx=out.idx[:ni].cpu().numpy().view(np.uint16),
               pos=out.pos[:nv].cpu().numpy(), color=out.color[:nv].cpu().numpy().view(np.uint32),
               meshes=out.meshes[:nm * 32].cpu().numpy().view(capi.mesh_dtype), uv=uv[:nv].cpu().numpy(), num_concave=nf)
    pset.close()
    return res
