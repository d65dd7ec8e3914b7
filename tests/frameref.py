"""Frame-level comparison against the reference's own Context (oracle/_ref/libvgref_vg.so): what vg::end() hands to bgfx
versus what vgx_cmdlist_decode + the tessellator + draw-command assembly produce. TEST INFRASTRUCTURE."""
import numpy as np

import pyvgref as R
import cmdlist_util as cu


def record(rc, script, flags=0):
    """Record `script` with the reference's own vg::clXxx writers; returns (handle, bytes)."""
    cl = rc.create_command_list(flags)
    script.play(rc, cl)
    data, ng, ni = rc.command_list_bytes(cl)
    return cl, data


def reference_frames(script, pres, canvas=(1280, 720), max_vb=65536, flags=0):
    """One command list submitted in len(pres) consecutive frames of ONE Context; pres[i] = Script played on the Context
    before the submission of frame i (the state the list is submitted under). Returns one dict per frame like
    reference_frame; with CommandListFlags::Cacheable the first frame populates the list's shape cache and the later
    ones render from it (clCacheRender, vg.cpp:5845-6135) as long as the average scale stays the same."""
    outs = []
    with R.RefContext(max_vb_vertices=max_vb) as rc:
        root, data = record(rc, script, flags)
        for pre in pres:
            rc.begin(canvas[0], canvas[1], 1.0)
            if pre is not None:
                pre.play(rc, R.IMMEDIATE)
            st0 = rc.state()
            rc.op(R.IMMEDIATE, R.SubmitCommandList, (), (root,))
            fr = rc.end()
            outs.append(dict(frame=fr, bytes=data, lists={root: (data, flags)}, root=root, params=rc.params(), state0=st0,
                             white_uv=rc.white_uv(), cache=rc.cache(root) if (flags & R.CL_CACHEABLE) else None))
            rc.next_frame()
    return outs


def reference_frame(script, canvas=(1280, 720), max_vb=65536, flags=0, children=(), immediate=False, pre=None, frames=1, uv_float=False, images=0, compat=False):
    """Play one frame on the reference. children: [(Script, flags)] recorded first (handles 0..), the root list after
    them. Returns dict(frame=Frame, bytes=root bytes, lists={handle: (bytes, flags)}, root=handle, params, state0)."""
    # uv_float: the VG_CONFIG_UV_INT16=0 build of the reference; compat: the reference's vg.cpp over the product's libvgx_compat.so
    with R.RefContext(max_vb_vertices=max_vb, uv_float=uv_float, compat=compat) as rc:
        lists = {}
        img = [rc.create_image(8, 8) for _ in range(images)]  # user images (handles after the font atlas) for IndexedTriList
        assert all(h != 0xFFFF for h in img)
        for cs, cf in children:
            h, b = record(rc, cs, cf)
            lists[h] = (b, cf)
        root, data = (None, b"")
        if not immediate:
            root, data = record(rc, script, flags)
            lists[root] = (data, flags)
        out = None
        for _ in range(frames):
            rc.begin(canvas[0], canvas[1], 1.0)
            if pre is not None:
                pre.play(rc, R.IMMEDIATE)
            st0 = rc.state()
            if immediate:
                script.play(rc, R.IMMEDIATE)
            else:
                rc.op(R.IMMEDIATE, R.SubmitCommandList, (), (root,))
            fr = rc.end()
            cache = rc.cache(root) if (not immediate and (flags & R.CL_CACHEABLE)) else None
            out = dict(frame=fr, bytes=data, lists=lists, root=root, params=rc.params(), state0=st0, white_uv=rc.white_uv(), cache=cache,
                       font_image=rc.font_image(), uv_float=uv_float)
            rc.next_frame()
        return out


def decode(rt, ref, canvas=(1280, 720), flags=0):
    """vgx_cmdlist_decode of the root list's bytes under the state the reference had at submission."""
    st0 = ref["state0"]
    extra = {}
    rc, ps, draws, n = cu.decode(rt, ref["bytes"], mtx=st0["mtx"].tolist(), global_alpha=st0["global_alpha"], tess_tol=ref["params"]["tess_tol"],
                                 fringe=ref["params"]["fringe"], canvas=(float(canvas[0]), float(canvas[1])), flags=flags,
                                 lists={h: v for h, v in ref["lists"].items() if h != ref["root"]}, extra=extra,
                                 white_uv=ref["white_uv"][0] if "white_uv" in ref else None, font_image=ref.get("font_image", 0), uv_float=ref.get("uv_float", False))
    assert rc == 0, rc
    return ps, draws, n, extra


def merged_reference_commands(fr):
    """The reference keeps draw commands and clip commands in two tables; both index the same vertex / index buffers.
    Merge them in buffer order (first_index is strictly increasing over the frame)."""
    cmds = [(int(c["first_index"]), 0, c) for c in fr.drawcmds] + [(int(c["first_index"]), 1, c) for c in fr.clipcmds]
    cmds.sort(key=lambda t: t[0])
    return [(c, bool(isclip)) for _, isclip, c in cmds]


def assert_frame_equal(fr, pos, color, idx, meshes, cmds, draws, dstate, max_vb, white_uv=None, uv=None, what=""):
    """fr: reference Frame. pos / color / idx / meshes: tessellator output with assembly applied (idx rebased),
    cmds: vgx_drawcmd records. Bit-exact: positions, indices, colours (where the reference writes them), commands."""
    rpos, rcol, ruv = R.frame_streams(fr)
    # vertex buffers: the reference fills each to <= max_vb and starts the next; our streams are their concatenation
    assert pos.shape[0] == rpos.shape[0], (what, pos.shape, rpos.shape)
    assert np.array_equal(pos.view(np.uint32), rpos.view(np.uint32)), (what, "pos")
    assert np.array_equal(idx, fr.idx), (what, "idx", np.flatnonzero(idx != fr.idx)[:5] if idx.shape == fr.idx.shape else (idx.shape, fr.idx.shape))
    ref_cmds = merged_reference_commands(fr)
    assert len(cmds) == len(ref_cmds), (what, len(cmds), len(ref_cmds))
    vb_first = np.concatenate([[0], np.cumsum([v["pos"].shape[0] for v in fr.vbs])])
    for i, (c, (rcmd, isclip)) in enumerate(zip(cmds, ref_cmds)):
        key = int(c["state_key"])
        typ, handle = (key >> 16) & 3, key & 0xFFFF
        assert typ == (3 if isclip else int(rcmd["type"])), (what, i, typ, rcmd)
        if typ in (1, 2):
            assert handle == int(rcmd["handle"]), (what, i, handle, rcmd)
        assert int(c["vertex_buffer"]) == int(rcmd["vertex_buffer"]), (what, i, c, rcmd)
        assert int(c["first_vertex_in_vb"]) == int(rcmd["first_vertex"]), (what, i, c, rcmd)
        assert int(c["first_vertex"]) == int(vb_first[int(rcmd["vertex_buffer"])]) + int(rcmd["first_vertex"]), (what, i)
        assert int(c["first_index"]) == int(rcmd["first_index"]), (what, i, c, rcmd)
        assert int(c["num_vertices"]) == int(rcmd["num_vertices"]) and int(c["num_indices"]) == int(rcmd["num_indices"]), (what, i, c, rcmd)
        m0 = int(c["first_mesh"])
        d0 = int(meshes["draw"][m0])
        if dstate is not None:
            assert dstate["scissor"][d0].tolist() == rcmd["scissor"].tolist(), (what, i, dstate["scissor"][d0], rcmd["scissor"])
            if not isclip:
                # clip region: the reference stores a range of clip COMMANDS, we a range of clip DRAWS; same region when
                # the commands of the one cover exactly the meshes of the other
                rf, rn = int(rcmd["clip_first_cmd"]), int(rcmd["clip_num_cmds"])
                gf, gn = int(dstate["clip_first_draw"][d0]), int(dstate["clip_num_draws"][d0])
                assert (rf == 0xFFFFFFFF) == (gf == 0xFFFFFFFF), (what, i, rf, gf)
                if rf != 0xFFFFFFFF:
                    assert int(dstate["clip_rule"][d0]) == int(rcmd["clip_rule"])
                    rv = sum(int(fr.clipcmds[k]["num_vertices"]) for k in range(rf, rf + rn))
                    is_clip_draw = ((draws["state_key"] >> 16) & 3) == 3  # the region = the Clip draws inside the range
                    sel = (meshes["draw"] >= gf) & (meshes["draw"] < gf + gn) & is_clip_draw[meshes["draw"]]
                    assert rv == int(meshes["num_vertices"][sel].sum()), (what, i, "clip region")
                    if rn:
                        fv = int(vb_first[int(fr.clipcmds[rf]["vertex_buffer"])]) + int(fr.clipcmds[rf]["first_vertex"])
                        assert fv == int(meshes["first_vertex"][sel][0]), (what, i, "clip region start")
        # colours: createDrawCommand_Clip writes none (the buffer keeps whatever it held), every other flavour does
        v0, v1 = int(c["first_vertex"]), int(c["first_vertex"]) + int(c["num_vertices"])
        if not isclip:
            assert np.array_equal(color[v0:v1], rcol[v0:v1]), (what, i, "color")
        # UVs: only createDrawCommand_VertexColor (Textured) writes the white-pixel UV
        if uv is not None and typ == 0:
            assert np.array_equal(uv[v0:v1], ruv[v0:v1]), (what, i, "uv")


def cpu_frame(oracle, ps, draws, max_vb):
    """The reference's path / stroker sources (oracle/_ref/libvgref.so) + the restated assembler, on the decoded batch."""
    res = oracle.tessellate(ps, draws)
    keys = draws["state_key"][res.meshes["draw"]] if res.meshes.shape[0] else np.zeros(0, np.uint32)
    st, cmds, idx = oracle.assemble(res.meshes, res.idx, max_vb, mesh_keys=keys)
    assert st == 0, st
    return res, cmds, idx


def cache_instances(capi, meshes, draws_now, dstate_now=None):
    """One vgx_cache_instance per draw (= per CachedCommand, vg.cpp:5773-5806): its mesh range in the cached drawing and the
    transform the state has when the fill / stroke command is replayed (submitCachedMesh, :6137-6166)."""
    n = draws_now.shape[0]
    inst = np.zeros(n, dtype=capi.cache_instance_dtype)
    d = meshes["draw"].astype(np.int64)
    first = np.searchsorted(d, np.arange(n), side="left")
    last = np.searchsorted(d, np.arange(n), side="right")
    inst["first_mesh"] = first
    inst["num_meshes"] = last - first
    inst["mtx"] = draws_now["mtx"]
    if dstate_now is not None:
        inst["color"] = dstate_now["raw_color"]  # what clCacheRender hands to submitCachedMesh: the command's Color operand
    return inst
