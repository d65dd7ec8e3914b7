"""ctypes loader for oracle/_ref/libvgref_vg.so: the reference's own src/vg.cpp (Context, draw-command assembly,
command lists, shape cache) compiled unmodified behind a recording bgfx stand-in.  TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module. It is the parity oracle of SURVEY 8(f)-1 (draw-command assembly,
vg.cpp:5207-5460 + what vg::end hands to bgfx, :1076-1288), 8(f)-2 (command-list byte-code: the reference's own
vg::clXxx writers produce the bytes vgx_cmdlist_decode reads) and 8(f)-3 (shape cache, :5773-6211).
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.environ.get("VGREF_VG_SO", os.path.join(_HERE, "_ref", "libvgref_vg.so"))
IMMEDIATE = 0xFFFFFFFF

# vg::CommandType::Enum (reference src/vg.cpp:177-241); the op codes of RefContext.op
(BeginPath, MoveTo, LineTo, CubicTo, QuadraticTo, ArcTo, Arc, Rect, RoundedRect, RoundedRectVarying, Circle, Ellipse, Polyline,
 ClosePath, FillPathColor, FillPathGradient, FillPathImagePattern, StrokePathColor, StrokePathGradient, StrokePathImagePattern,
 IndexedTriList, BeginClip, EndClip, ResetClip, CreateLinearGradient, CreateBoxGradient, CreateRadialGradient, CreateImagePattern,
 PushState, PopState, ResetScissor, SetScissor, IntersectScissor, TransformIdentity, TransformScale, TransformTranslate,
 TransformRotate, TransformMult, SetViewBox, SetGlobalAlpha, Text, TextBox, SubmitCommandList) = range(43)

# include/vg/vg.h:156-259
def stroke_flags(cap, join, aa, fixed_width=False):
    return (int(aa) << 4) | (cap << 2) | join | ((1 << 5) if fixed_width else 0)


def fill_flags(aa, concave=False, even_odd=False):
    return (int(even_odd) << 4) | (int(aa) << 2) | int(concave)


CL_CACHEABLE = 1
CL_ALLOW_CULLING = 2

drawcmd_dtype = np.dtype([("type", "<u4"), ("vertex_buffer", "<u4"), ("first_vertex", "<u4"), ("first_index", "<u4"),
                          ("num_vertices", "<u4"), ("num_indices", "<u4"), ("scissor", "<u2", (4,)), ("handle", "<u4"),
                          ("clip_rule", "<u4"), ("clip_first_cmd", "<u4"), ("clip_num_cmds", "<u4")], align=True)
submit_dtype = np.dtype([("program", "<u4"), ("vb_pos", "<u4"), ("first_vertex", "<u4"), ("num_vertices", "<u4"),
                         ("first_index", "<u4"), ("num_indices", "<u4"), ("has_color_stream", "<u4"), ("has_uv_stream", "<u4"),
                         ("scissor", "<u2", (4,)), ("stencil", "<u4"), ("texture", "<u4"), ("state", "<u8"),
                         ("num_uniforms", "<u4"), ("reserved", "<u4"), ("paint_mat", "<f4", (9,)), ("params", "<f4", (4,)),
                         ("inner_color", "<f4", (4,)), ("outer_color", "<f4", (4,))], align=True)
assert drawcmd_dtype.itemsize == 48 and submit_dtype.itemsize == 152  # sizeof(vgr_drawcmd), sizeof(vgr_submit)

_lib = None


def available():
    return os.path.exists(PATH)


PATH_UV_FLOAT = os.path.join(_HERE, "_ref", "libvgref_vg_uvf.so")  # the same sources with -DVG_CONFIG_UV_INT16=0 (float UVs)
# the reference's Context with its path.cpp / stroker.cpp replaced by the product's libvgx_compat.so (oracle/Makefile): the
# drop-in check of SURVEY 8(b), played by tests/test_compat_context.py
PATH_COMPAT = os.path.join(_HERE, "_ref", "libvgref_vg_compat.so")
_libs = {}


def load(path=None):
    global _lib
    path = path or PATH
    if path in _libs:
        return _libs[path]
    lib = C.CDLL(path)
    lib.vgr_create.restype = C.c_void_p
    lib.vgr_create.argtypes = [C.c_uint32] * 4
    lib.vgr_destroy.argtypes = [C.c_void_p]
    lib.vgr_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float]
    lib.vgr_end.argtypes = [C.c_void_p]
    lib.vgr_frame.argtypes = [C.c_void_p]
    lib.vgr_cl_create.restype = C.c_uint32
    lib.vgr_cl_create.argtypes = [C.c_void_p, C.c_uint32]
    lib.vgr_cl_destroy.argtypes = [C.c_void_p, C.c_uint32]
    lib.vgr_cl_reset.argtypes = [C.c_void_p, C.c_uint32]
    lib.vgr_create_image.restype = C.c_uint32
    lib.vgr_create_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.vgr_cl_bytes.restype = C.c_int
    lib.vgr_cl_bytes.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.vgr_op.restype = C.c_uint32
    lib.vgr_op.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.vgr_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vgr_get_params.argtypes = [C.c_void_p, C.c_void_p]
    lib.vgr_white_uv.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    lib.vgr_font_image.restype = C.c_uint32
    lib.vgr_font_image.argtypes = [C.c_void_p]
    lib.vgr_num_vertex_buffers.restype = C.c_uint32
    lib.vgr_num_vertex_buffers.argtypes = [C.c_void_p]
    lib.vgr_vertex_buffer.restype = C.c_int
    lib.vgr_vertex_buffer.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.vgr_index_buffer.restype = C.c_int
    lib.vgr_index_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    for f in (lib.vgr_draw_commands, lib.vgr_clip_commands, lib.vgr_submits):
        f.restype = C.c_uint32
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.vgr_cache_info.restype = C.c_int
    lib.vgr_cache_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    lib.vgr_cache_mesh.restype = C.c_int
    lib.vgr_cache_mesh.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.vgr_cache_command.restype = C.c_int
    lib.vgr_cache_command.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p]
    lib.vgr_engine_name.restype = C.c_char_p
    _libs[path] = lib
    if path == PATH:
        _lib = lib
    return lib


def _bytes_at(ptr, n):
    if not ptr or n == 0:
        return b""
    return C.string_at(ptr, n)


class Frame:
    """What vg::end() handed to bgfx for one frame."""
    pass


class RefContext:
    """One vg::Context of the reference. `cl` arguments: IMMEDIATE plays a call on the Context (vg::xxx), a command
    list handle records it with the reference's own vg::clXxx writer."""

    def __init__(self, max_vb_vertices=65536, max_command_lists=256, max_gradients=64, max_image_patterns=64, uv_float=False, compat=False):
        self.lib = load(PATH_COMPAT if compat else (PATH_UV_FLOAT if uv_float else None))
        self.uv_dtype = np.float32 if uv_float else np.int16
        self.h = self.lib.vgr_create(max_vb_vertices, max_command_lists, max_gradients, max_image_patterns)

    def close(self):
        if self.h:
            self.lib.vgr_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def begin(self, w=1280, h=720, dpr=1.0):
        self.lib.vgr_begin(self.h, w, h, dpr)

    def op(self, cl, code, f=(), u=()):
        fa = np.ascontiguousarray(f, dtype=np.float32).reshape(-1)
        ua = np.ascontiguousarray(u, dtype=np.uint32).reshape(-1)
        return self.lib.vgr_op(self.h, cl, code, fa.ctypes.data if fa.size else None, ua.ctypes.data if ua.size else None)

    def create_command_list(self, flags=0):
        return self.lib.vgr_cl_create(self.h, flags)

    def destroy_command_list(self, cl):
        self.lib.vgr_cl_destroy(self.h, cl)

    def reset_command_list(self, cl):
        self.lib.vgr_cl_reset(self.h, cl)

    def create_image(self, w=64, h=64, flags=0):
        return self.lib.vgr_create_image(self.h, w, h, flags)

    def command_list_bytes(self, cl):
        """CommandList::m_CommandBuffer[0 .. m_CommandBufferPos) as bytes + (numGradients, numImagePatterns)."""
        p = C.c_void_p()
        n = C.c_uint32()
        ng = C.c_uint32()
        ni = C.c_uint32()
        rc = self.lib.vgr_cl_bytes(self.h, cl, C.byref(p), C.byref(n), C.byref(ng), C.byref(ni))
        assert rc == 0, rc
        return _bytes_at(p.value, n.value), ng.value, ni.value

    def state(self):
        m = np.zeros(6, np.float32)
        s = np.zeros(4, np.float32)
        a = np.zeros(3, np.float32)
        self.lib.vgr_get_state(self.h, m.ctypes.data, s.ctypes.data, a.ctypes.data)
        return dict(mtx=m, scissor=s, global_alpha=float(a[0]), avg_scale=float(a[1]), font_scale=float(a[2]))

    def params(self):
        p = np.zeros(2, np.float32)
        self.lib.vgr_get_params(self.h, p.ctypes.data)
        return dict(tess_tol=float(p[0]), fringe=float(p[1]))

    def font_image(self):
        return int(self.lib.vgr_font_image(self.h))

    def white_uv(self):
        raw = np.zeros(2, np.uint32)
        nb = C.c_uint32()
        self.lib.vgr_white_uv(self.h, raw.ctypes.data, C.byref(nb))
        return raw, nb.value

    def end(self):
        """vg::end(); returns the captured Frame: per vertex buffer pos / uv / color arrays, the index buffer, the
        Context's draw / clip command tables and the bgfx submits."""
        self.lib.vgr_end(self.h)
        fr = Frame()
        nvb = self.lib.vgr_num_vertex_buffers(self.h)
        fr.vbs = []
        p = C.c_void_p()
        n = C.c_uint64()
        cmds = np.zeros(1, dtype=drawcmd_dtype)
        ncmd = self.lib.vgr_draw_commands(self.h, cmds.ctypes.data, 0)
        cmds = np.zeros(max(ncmd, 1), dtype=drawcmd_dtype)
        self.lib.vgr_draw_commands(self.h, cmds.ctypes.data, ncmd)
        fr.drawcmds = cmds[:ncmd]
        nclip = self.lib.vgr_clip_commands(self.h, cmds.ctypes.data, 0)
        clips = np.zeros(max(nclip, 1), dtype=drawcmd_dtype)
        self.lib.vgr_clip_commands(self.h, clips.ctypes.data, nclip)
        fr.clipcmds = clips[:nclip]
        if ncmd == 0:
            fr.idx = np.zeros(0, np.uint16)
            fr.submits = np.zeros(0, dtype=submit_dtype)
            return fr
        for i in range(nvb):
            vb = {}
            for s, (name, dt, w) in enumerate((("pos", np.float32, 2), ("uv", self.uv_dtype, 2), ("color", np.uint32, 1))):
                rc = self.lib.vgr_vertex_buffer(self.h, i, s, C.byref(p), C.byref(n))
                assert rc == 0, (rc, i, s)
                a = np.frombuffer(_bytes_at(p.value, n.value), dtype=dt)
                vb[name] = a.reshape(-1, w) if w > 1 else a
            fr.vbs.append(vb)
        rc = self.lib.vgr_index_buffer(self.h, C.byref(p), C.byref(n))
        assert rc == 0, rc
        fr.idx = np.frombuffer(_bytes_at(p.value, n.value), dtype=np.uint16)
        ns = self.lib.vgr_submits(self.h, None, 0)
        sub = np.zeros(max(ns, 1), dtype=submit_dtype)
        self.lib.vgr_submits(self.h, sub.ctypes.data, ns)
        fr.submits = sub[:ns]
        return fr

    def next_frame(self):
        self.lib.vgr_frame(self.h)

    def cache(self, cl):
        """CommandListCache of a Cacheable list: dict(avg_scale, meshes=[(pos, colors|None, idx)], commands=[(first, n, inv)])
        or None when the list has no cache yet."""
        nm = C.c_uint32()
        nc = C.c_uint32()
        sc = C.c_float()
        if self.lib.vgr_cache_info(self.h, cl, C.byref(nm), C.byref(nc), C.byref(sc)) != 0:
            return None
        meshes = []
        pp, pc, pi = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nv, ni = C.c_uint32(), C.c_uint32()
        for i in range(nm.value):
            assert self.lib.vgr_cache_mesh(self.h, cl, i, C.byref(pp), C.byref(pc), C.byref(pi), C.byref(nv), C.byref(ni)) == 0
            pos = np.frombuffer(_bytes_at(pp.value, nv.value * 8), dtype=np.float32).reshape(-1, 2)
            col = np.frombuffer(_bytes_at(pc.value, nv.value * 4), dtype=np.uint32) if pc.value else None
            idx = np.frombuffer(_bytes_at(pi.value, ni.value * 2), dtype=np.uint16)
            meshes.append((pos, col, idx))
        cmds = []
        fm, nmm = C.c_uint32(), C.c_uint32()
        inv = np.zeros(6, np.float32)
        for i in range(nc.value):
            assert self.lib.vgr_cache_command(self.h, cl, i, C.byref(fm), C.byref(nmm), inv.ctypes.data) == 0
            cmds.append((fm.value, nmm.value, inv.copy()))
        return dict(avg_scale=sc.value, meshes=meshes, commands=cmds)


# ---- helpers shared by the tests ------------------------------------------------------------------------------------------
_PATH_OPS = {0: MoveTo, 1: LineTo, 2: CubicTo, 3: QuadraticTo, 4: ClosePath, 5: ArcTo, 6: Arc, 7: Rect, 8: RoundedRect,
             9: RoundedRectVarying, 10: Circle, 11: Ellipse, 12: Polyline}  # vgx_cmd -> vg::CommandType


def play_path(rc, cl, ps, p):
    """Issue path p of a PathSetArrays through the reference (BeginPath + its commands)."""
    rc.op(cl, BeginPath)
    for k in range(int(ps.path_cmd_begin[p]), int(ps.path_cmd_begin[p + 1])):
        t = int(ps.cmd_type[k])
        a = ps.args[int(ps.cmd_arg_off[k]):int(ps.cmd_arg_off[k + 1])]
        if t == 6:
            rc.op(cl, Arc, a[:5], [1 if a[5] != 0 else 0])
        elif t == 12:
            rc.op(cl, Polyline, a, [len(a) // 2])
        else:
            rc.op(cl, _PATH_OPS[t], a)


def frame_streams(fr):
    """Concatenate a Frame's vertex buffers (the order the frame's vertices were appended in)."""
    if not fr.vbs:
        z = np.zeros((0, 2), np.float32)
        return z, np.zeros(0, np.uint32), np.zeros((0, 2), np.int16)
    return (np.concatenate([v["pos"] for v in fr.vbs]), np.concatenate([v["color"] for v in fr.vbs]),
            np.concatenate([v["uv"] for v in fr.vbs]))
