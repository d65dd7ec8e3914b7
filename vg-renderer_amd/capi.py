"""ctypes/numpy mirror of include/vgx.h (the C-ABI of libvgx.so).

Plumbing only: struct layouts, enums and the function prototypes. The same struct layouts are used by
the CPU oracles under oracle/ (they export vgo_* entry points with host pointers).
"""
import ctypes as C
import numpy as np

# ---- enums (include/vgx.h) -------------------------------------------------------------------
VGX_OK = 0
VGX_E_INVALID_ARG = 1
VGX_E_INVALID_PATH = 2
VGX_E_NONFINITE = 3
VGX_E_NOSPACE = 4
VGX_E_MESH_TOO_LARGE = 5
VGX_E_HIP = 6
VGX_E_NO_DEVICE = 7
VGX_E_RANGE = 8
VGX_E_INTERNAL = 9
VGX_E_STALE = 10
FILL_TRILIST = 0x40  # vgx_draw.fill_flags: a user mesh (IndexedTriList): the decoder's tri_* arrays + vgx_merge_uv
FILL_CONCAVE, FILL_EVEN_ODD = 0x10, 0x20  # vgx_draw.fill_flags: a concave fill (no GPU mesh: libtess2 + vgx_concave_* + vgx_merge)

CMD_MOVE_TO, CMD_LINE_TO, CMD_CUBIC_TO, CMD_QUAD_TO, CMD_CLOSE = 0, 1, 2, 3, 4
CMD_ARC_TO, CMD_ARC, CMD_RECT, CMD_ROUNDED_RECT, CMD_ROUNDED_RECT_VARYING = 5, 6, 7, 8, 9
CMD_CIRCLE, CMD_ELLIPSE, CMD_POLYLINE = 10, 11, 12
CMD_ARG_COUNT = [2, 2, 6, 4, 0, 5, 6, 4, 5, 8, 3, 4, -1]

CAP_BUTT, CAP_ROUND, CAP_SQUARE = 0, 1, 2
JOIN_MITER, JOIN_ROUND, JOIN_BEVEL = 0, 1, 2

FILL_ENABLE, FILL_AA = 0x1, 0x2
FILL_INDEX_ORDER_SSE = 0x100  # strokerConvexFillAA indices in the order of the reference's SSE2 variant (stroker.cpp:610-701)
STROKE_ENABLE, STROKE_AA, STROKE_THIN = 0x1, 0x2, 0x4

MESH_FILL, MESH_FILL_AA, MESH_STROKE, MESH_STROKE_AA, MESH_STROKE_AA_THIN, MESH_CONCAVE_FILL_AA, MESH_TRILIST = 0, 1, 2, 3, 4, 5, 6


def stroke_flags(cap, join, aa=True, thin=False):
    return STROKE_ENABLE | (STROKE_AA if aa else 0) | (STROKE_THIN if thin else 0) | (cap << 4) | (join << 6)


def fill_flags(aa=True):
    return FILL_ENABLE | (FILL_AA if aa else 0)


# ---- numpy record layouts ---------------------------------------------------------------------
draw_dtype = np.dtype([
    ("path", "<u4"), ("fill_flags", "<u4"), ("fill_color", "<u4"), ("stroke_flags", "<u4"),
    ("stroke_color", "<u4"), ("stroke_width", "<f4"), ("scale", "<f4"), ("tess_tol", "<f4"),
    ("fringe", "<f4"), ("mtx", "<f4", (6,)), ("state_key", "<u4")])
assert draw_dtype.itemsize == 64

subpath_dtype = np.dtype([("first_vertex", "<u8"), ("num_vertices", "<u4"), ("flags", "<u4")])
assert subpath_dtype.itemsize == 16

draw_info_dtype = np.dtype([
    ("first_poly_vertex", "<u8"), ("first_subpath", "<u8"), ("first_mesh", "<u8"),
    ("num_poly_vertices", "<u4"), ("num_subpaths", "<u4"), ("num_meshes", "<u4"), ("flags", "<u4")])
assert draw_info_dtype.itemsize == 40

mesh_dtype = np.dtype([
    ("first_vertex", "<u8"), ("first_index", "<u8"), ("num_vertices", "<u4"), ("num_indices", "<u4"),
    ("draw", "<u4"), ("subpath_kind", "<u4")])
assert mesh_dtype.itemsize == 32
contour_dtype = np.dtype([("first_vertex", "<u8"), ("num_vertices", "<u4"), ("fill", "<u4")])
assert contour_dtype.itemsize == 16
concave_fill_dtype = np.dtype([("first_contour", "<u8"), ("num_contours", "<u4"), ("color", "<u4"), ("fringe", "<f4"),
                               ("num_tess_vertices", "<u4"), ("num_tess_indices", "<u4"), ("reserved", "<u4"),
                               ("first_tess_vertex", "<u8"), ("first_tess_index", "<u8")])
assert concave_fill_dtype.itemsize == 48
drawcmd_dtype = np.dtype([
    ("first_vertex", "<u8"), ("first_index", "<u8"), ("first_mesh", "<u8"), ("num_vertices", "<u4"), ("num_indices", "<u4"),
    ("num_meshes", "<u4"), ("vertex_buffer", "<u4"), ("first_vertex_in_vb", "<u4"), ("state_key", "<u4")])
assert drawcmd_dtype.itemsize == 48
ASM_SPLIT_STATE = 1


# ---- ctypes structs ---------------------------------------------------------------------------
class Sizes(C.Structure):
    _fields_ = [("num_poly_vertices", C.c_uint64), ("num_subpaths", C.c_uint64), ("num_meshes", C.c_uint64),
                ("num_vertices", C.c_uint64), ("num_indices", C.c_uint64), ("num_serial_draws", C.c_uint64),
                ("num_cmd_instances", C.c_uint64), ("num_elements", C.c_uint64), ("num_fill_elements", C.c_uint64),
                ("num_drawcmds", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class PathSetDesc(C.Structure):
    _fields_ = [("cmd_type", C.c_void_p), ("cmd_arg_off", C.c_void_p), ("args", C.c_void_p),
                ("path_cmd_begin", C.c_void_p), ("npaths", C.c_uint32), ("ncmd", C.c_uint32)]


class FlatOut(C.Structure):
    _fields_ = [("poly", C.c_void_p), ("subpaths", C.c_void_p), ("draw_info", C.c_void_p),
                ("cap_poly_vertices", C.c_uint64), ("cap_subpaths", C.c_uint64)]


cache_instance_dtype = np.dtype([("first_mesh", "<u8"), ("num_meshes", "<u4"), ("color", "<u4"), ("mtx", "<f4", (6,))])
assert cache_instance_dtype.itemsize == 40


class CacheDesc(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("color", C.c_void_p), ("idx", C.c_void_p), ("meshes", C.c_void_p),
                ("num_meshes", C.c_uint64), ("num_vertices", C.c_uint64), ("num_indices", C.c_uint64)]


class Assembly(C.Structure):
    _fields_ = [("drawcmds", C.c_void_p), ("cap_drawcmds", C.c_uint64), ("dev_num_drawcmds", C.c_void_p),
                ("max_vb_vertices", C.c_uint32), ("flags", C.c_uint32), ("uv", C.c_void_p), ("uv_bytes", C.c_uint32),
                ("uv_value", C.c_uint32 * 2), ("reserved", C.c_uint32)]


class MeshOut(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("color", C.c_void_p), ("idx", C.c_void_p), ("meshes", C.c_void_p),
                ("cap_vertices", C.c_uint64), ("cap_indices", C.c_uint64), ("cap_meshes", C.c_uint64)]


VGX_MAX_STAGES = 16


class StageTimes(C.Structure):
    _fields_ = [("num_stages", C.c_uint32), ("ms", C.c_float * VGX_MAX_STAGES), ("name", C.c_char_p * VGX_MAX_STAGES)]


class CmdListRef(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("size", C.c_uint32), ("flags", C.c_uint32)]


class CmdListState(C.Structure):
    _fields_ = [("mtx", C.c_float * 6), ("global_alpha", C.c_float), ("tess_tol", C.c_float), ("fringe", C.c_float),
                ("canvas_width", C.c_float), ("canvas_height", C.c_float), ("flags", C.c_uint32),
                ("scissor", C.c_float * 4), ("first_gradient", C.c_uint32), ("first_image_pattern", C.c_uint32),
                ("max_gradients", C.c_uint32), ("max_image_patterns", C.c_uint32), ("max_depth", C.c_uint32),
                ("num_lists", C.c_uint32), ("lists", C.POINTER(CmdListRef)), ("prev_cmd_scissor", C.c_uint16 * 4),
                ("prev_cmd_valid", C.c_uint32), ("first_generation", C.c_uint32),
                ("clip_valid", C.c_uint32), ("clip_rule", C.c_uint32), ("clip_first_draw", C.c_uint32), ("clip_num_draws", C.c_uint32),
                ("clip_recording", C.c_uint32), ("draw_base", C.c_uint32), ("white_uv", C.c_uint32 * 2), ("font_image", C.c_uint32)]


class CmdListOut(C.Structure):
    _fields_ = [("cmd_type", C.c_void_p), ("cmd_arg_off", C.c_void_p), ("args", C.c_void_p), ("path_cmd_begin", C.c_void_p), ("draws", C.c_void_p),
                ("draw_state", C.c_void_p), ("paints", C.c_void_p),
                ("cap_cmds", C.c_uint32), ("cap_args", C.c_uint32), ("cap_paths", C.c_uint32), ("cap_draws", C.c_uint32), ("cap_paints", C.c_uint32),
                ("num_cmds", C.c_uint32), ("num_args", C.c_uint32), ("num_paths", C.c_uint32), ("num_draws", C.c_uint32), ("num_paints", C.c_uint32),
                ("num_skipped", C.c_uint32), ("next_gradient", C.c_uint32), ("next_image_pattern", C.c_uint32), ("next_generation", C.c_uint32),
                ("end_mtx", C.c_float * 6), ("end_global_alpha", C.c_float),
                ("end_clip_valid", C.c_uint32), ("end_clip_rule", C.c_uint32), ("end_clip_first_draw", C.c_uint32), ("end_clip_num_draws", C.c_uint32),
                ("end_clip_recording", C.c_uint32), ("end_scissor", C.c_float * 4), ("reserved", C.c_uint32),
                ("tri_pos", C.c_void_p), ("tri_color", C.c_void_p), ("tri_uv", C.c_void_p), ("tri_idx", C.c_void_p), ("tri_meshes", C.c_void_p),
                ("cap_tri_vertices", C.c_uint32), ("cap_tri_indices", C.c_uint32), ("cap_tri_meshes", C.c_uint32),
                ("num_tri_vertices", C.c_uint32), ("num_tri_indices", C.c_uint32), ("num_tri_meshes", C.c_uint32)]


draw_state_dtype = np.dtype([("scissor", "<u2", (4,)), ("clip_rule", "<u4"), ("clip_first_draw", "<u4"), ("clip_num_draws", "<u4"), ("raw_color", "<u4")])
paint_dtype = np.dtype([("type", "<u4"), ("handle", "<u4"), ("matrix", "<f4", (9,)), ("params", "<f4", (4,)), ("inner_color", "<f4", (4,)),
                        ("outer_color", "<f4", (4,)), ("image", "<u4")])
assert draw_state_dtype.itemsize == 24 and paint_dtype.itemsize == 96
CL_CACHEABLE, CL_ALLOW_CULLING = 1, 2
CL_SCISSOR_SET, CL_UV_FLOAT = 0x100, 0x200  # vgx_cmdlist_state.flags (host-side switches, not CommandListFlags)


class FailureInfo(C.Structure):
    _fields_ = [("status", C.c_uint32), ("reason", C.c_uint32), ("aux", C.c_uint32), ("segment_items", C.c_uint32), ("segment", C.c_uint64), ("prof", C.c_uint64 * 16)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "prof"}
        d["prof"] = [int(x) for x in self.prof]
        return d


class RankSizes(C.Structure):
    _fields_ = [("num_vertices", C.c_uint64), ("num_indices", C.c_uint64), ("num_meshes", C.c_uint64), ("num_draws", C.c_uint64)]


# Every symbol include/vgx.h declares, with (restype, argtypes). tests/test_capi_symbols.py checks the
# built library exports all of them.
VGX_SYMBOLS = {
    "vgx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "vgx_destroy": (C.c_int, [C.c_void_p]),
    "vgx_set_assembly": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vgx_cache_localize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "vgx_merge": (C.c_int, [C.c_void_p, C.POINTER(CacheDesc), C.POINTER(CacheDesc), C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(MeshOut), C.c_void_p, C.c_void_p, C.c_void_p]),
    "vgx_merge_uv": (C.c_int, [C.c_void_p, C.POINTER(CacheDesc), C.POINTER(CacheDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(MeshOut), C.c_void_p, C.c_void_p, C.c_void_p]),
    "vgx_cache_submit": (C.c_int, [C.c_void_p, C.POINTER(CacheDesc), C.c_void_p, C.c_uint64, C.POINTER(MeshOut), C.c_void_p, C.c_void_p, C.c_void_p]),
    "vgx_last_hip_error": (C.c_int, [C.c_void_p]),
    "vgx_status_string": (C.c_char_p, [C.c_int]),
    "vgx_version": (C.c_uint32, []),
    "vgx_scratch_bytes": (C.c_uint64, [C.c_void_p]),
    "vgx_pathset_create": (C.c_int, [C.c_void_p, C.POINTER(PathSetDesc), C.POINTER(C.c_void_p)]),
    "vgx_partition": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]),
    "vgx_pathset_validate": (C.c_int, [C.POINTER(PathSetDesc)]),
    "vgx_pathset_read_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vgx_pathset_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vgx_flatten_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Sizes), C.c_void_p]),
    "vgx_flatten_emit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(FlatOut), C.c_void_p]),
    "vgx_flatten": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(FlatOut), C.c_void_p, C.c_void_p, C.c_void_p]),
    "vgx_tessellate_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Sizes), C.c_void_p]),
    "vgx_tessellate_emit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(MeshOut), C.c_void_p]),
    "vgx_tessellate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(MeshOut), C.c_void_p, C.c_void_p, C.c_void_p]),
    "vgx_stroke_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(Sizes), C.c_void_p]),
    "vgx_stroke_emit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(MeshOut), C.c_void_p]),
    "vgx_concave_move": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "vgx_concave_emit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                   C.POINTER(MeshOut), C.c_void_p, C.c_void_p, C.c_void_p]),
    "vgx_cmdlist_decode": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(CmdListState), C.POINTER(CmdListOut)]),
    "vgx_get_failure_info": (C.c_int, [C.c_void_p, C.POINTER(FailureInfo), C.c_void_p]),
    "vgx_gather_sizes": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RankSizes), C.POINTER(RankSizes), C.c_void_p]),
    "vgx_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(MeshOut), C.POINTER(RankSizes), C.POINTER(MeshOut), C.c_void_p]),
    "vgx_gather_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(MeshOut), C.POINTER(RankSizes), C.POINTER(RankSizes), C.POINTER(MeshOut), C.c_void_p]),
    "vgx_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "vgx_set_static_batches": (C.c_int, [C.c_void_p, C.c_int]),
    "vgx_get_stage_times_avg": (C.c_int, [C.c_void_p, C.POINTER(StageTimes), C.c_uint32]),
    "vgx_get_stage_times": (C.c_int, [C.c_void_p, C.POINTER(StageTimes)]),
}


def bind(lib, symbols):
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
