"""Python plumbing over vgx_cmdlist_decode (include/vgx.h): a recorded vg-renderer command list (bytes as the reference's
vg::clXxx writers produce them, src/vg.cpp:2403-2690) -> path set + draw records, under the state the list is submitted in.
Two calls of the C entry point: a count pass, then the store pass into numpy arrays. Plumbing only."""
import numpy as np


def decode(rt, data, mtx=(1, 0, 0, 1, 0, 0), global_alpha=1.0, tess_tol=0.25, fringe=1.0, canvas=(1280.0, 720.0), flags=0,
           lists=None, first_gradient=0, first_image_pattern=0, extra=None, scissor=None, prev_cmd_scissor=None, first_generation=0, clip=None, draw_base=0,
           white_uv=None, font_image=0, uv_float=False, time_reps=0):
    """vgx_cmdlist_decode, count pass + store pass. Returns (status, PathSetArrays or None, draws ndarray, info dict).
    lists: {handle: (bytes, flags)} for SubmitCommandList. extra: dict that receives draw_state / paints / the out struct."""
    import ctypes as C
    import importlib
    capi = rt.capi
    pathset = importlib.import_module("vg-renderer_amd.pathset")
    st = capi.CmdListState()
    for i in range(6):
        st.mtx[i] = mtx[i]
    st.global_alpha = global_alpha; st.tess_tol = tess_tol; st.fringe = fringe
    st.canvas_width, st.canvas_height = canvas
    st.flags = flags
    st.first_gradient = first_gradient; st.first_image_pattern = first_image_pattern
    if scissor is not None:            # State::m_ScissorRect at submission: an explicit rectangle, possibly empty (VGX_CL_SCISSOR_SET)
        st.flags = flags | 0x100
        for i in range(4):
            st.scissor[i] = float(scissor[i])
    if prev_cmd_scissor is not None:   # scissor of the frame's last draw command so far (PopState rule, vg.cpp:3950-3965)
        for i in range(4):
            st.prev_cmd_scissor[i] = int(prev_cmd_scissor[i])
        st.prev_cmd_valid = 1
    st.first_generation = first_generation
    st.draw_base = draw_base
    st.font_image = font_image         # ctx->m_FontImages[0].idx: the image of colour draws and of IndexedTriList without one
    if white_uv is not None:           # getWhitePixelUV: the UV of tri-list vertices that come without UVs
        st.white_uv[0], st.white_uv[1] = int(white_uv[0]), int(white_uv[1])
    if uv_float:
        st.flags |= 0x200              # VGX_CL_UV_FLOAT: uv_t = float (VG_CONFIG_UV_INT16 = 0)
    if clip is not None:               # (valid, rule, first draw, draws, recording): the previous decode's end_clip_*
        st.clip_valid, st.clip_rule, st.clip_first_draw, st.clip_num_draws, st.clip_recording = [int(x) for x in clip]
    keep = []
    if lists:
        n = max(lists) + 1
        arr = (capi.CmdListRef * n)()
        for h, (b, fl) in lists.items():
            cb = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if len(b) else b"\0")
            keep.append(cb)
            arr[h].bytes = C.cast(cb, C.c_void_p); arr[h].size = len(b); arr[h].flags = fl
        st.lists = arr; st.num_lists = n
    out = capi.CmdListOut()
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if len(data) else b"\0")
    rc = rt.lib().vgx_cmdlist_decode(buf, len(data), C.byref(st), C.byref(out))
    if rc != 0:
        return rc, None, None, None
    n = {k: int(getattr(out, "num_" + k)) for k in ("cmds", "args", "paths", "draws", "skipped")}
    npaints = int(out.num_paints)
    cmd_type = np.zeros(max(n["cmds"], 1), np.uint8)
    arg_off = np.zeros(n["cmds"] + 1, np.uint32)
    args = np.zeros(max(n["args"], 1), np.float32)
    pcb = np.zeros(n["paths"] + 1, np.uint32)
    draws = np.zeros(max(n["draws"], 1), capi.draw_dtype)
    dstate = np.zeros(max(n["draws"], 1), capi.draw_state_dtype)
    paints = np.zeros(max(npaints, 1), capi.paint_dtype)
    out.cmd_type, out.cmd_arg_off, out.args, out.path_cmd_begin, out.draws = (cmd_type.ctypes.data, arg_off.ctypes.data, args.ctypes.data, pcb.ctypes.data, draws.ctypes.data)
    out.draw_state, out.paints = dstate.ctypes.data, paints.ctypes.data
    out.cap_cmds, out.cap_args, out.cap_paths, out.cap_draws, out.cap_paints = n["cmds"], n["args"], n["paths"], n["draws"], npaints
    tv, ti, tm = int(out.num_tri_vertices), int(out.num_tri_indices), int(out.num_tri_meshes)
    tri = dict(pos=np.zeros((max(tv, 1), 2), np.float32), color=np.zeros(max(tv, 1), np.uint32),
               uv=np.zeros((max(tv, 1), 2), np.float32 if uv_float else np.int16), idx=np.zeros(max(ti, 1), np.uint16),
               meshes=np.zeros(max(tm, 1), capi.mesh_dtype))
    if tm:
        out.tri_pos, out.tri_color, out.tri_uv, out.tri_idx, out.tri_meshes = (tri[k].ctypes.data for k in ("pos", "color", "uv", "idx", "meshes"))
        out.cap_tri_vertices, out.cap_tri_indices, out.cap_tri_meshes = tv, ti, tm
    rc = rt.lib().vgx_cmdlist_decode(buf, len(data), C.byref(st), C.byref(out))
    if time_reps and extra is not None:  # the store pass again, timed (the C call alone: what a host pays per frame for the decode)
        import time
        t0 = time.perf_counter()
        for _ in range(int(time_reps)):
            rt.lib().vgx_cmdlist_decode(buf, len(data), C.byref(st), C.byref(out))
        extra["decode_seconds"] = (time.perf_counter() - t0) / int(time_reps)
    ps = pathset.PathSetArrays(cmd_type[:n["cmds"]], arg_off, args[:n["args"]], pcb)
    if extra is not None:
        extra["draw_state"] = dstate[:n["draws"]]
        extra["paints"] = paints[:npaints]
        extra["out"] = out
        extra["tri"] = dict(pos=tri["pos"][:tv], color=tri["color"][:tv], uv=tri["uv"][:tv], idx=tri["idx"][:ti], meshes=tri["meshes"][:tm])
    return rc, ps, draws[:n["draws"]], n
