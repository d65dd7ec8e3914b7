/*
 * vgx.h -- C-ABI of the MI355X-native batch geometry path (libvgx.so).
 *
 * This is the drop-in boundary for vg-renderer's CPU geometry hot path:
 *   vg::Path     bezier flattener        (reference include/vg/path.h:19-38,    src/path.cpp)
 *   vg::Stroker  stroke/fill/AA mesher   (reference include/vg/stroker.h:11-85, src/stroker.cpp)
 * The reference calls those once per path per frame (src/vg.cpp:2969-3059, 3061-3179, 3401-3492);
 * this ABI takes MANY path instances ("draws") at once so that one launch sequence on a gfx950
 * device does the work of millions of pathXXX/strokerXXX calls. Everything is plain pointers and
 * sizes; no C++ or torch types cross the boundary. The C++ header include/vgx_compat.hpp layers the
 * reference's own vg::pathXXX / vg::strokerXXX names on top of this ABI.
 *
 * Conventions
 *   - "host" pointers are ordinary CPU memory, "device" pointers are HIP device memory on the
 *     context's device. Each parameter says which one it is.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream). Calls are asynchronous
 *     unless they return data to the host (documented per function).
 *   - All functions return a vgx_status; they never abort and never print.
 *   - Numeric contract: IEEE binary32, no FMA contraction, transcendentals from csrc/vgmath.h.
 *     Indices are mesh-local uint16 (reference vg::Mesh, include/vg/vg.h:353-360), colours are
 *     uint32 0xAABBGGRR (include/vg/vg.h:80-86).
 */
#ifndef VGX_H
#define VGX_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGX_VERSION 1

typedef enum vgx_status {
	VGX_OK = 0,
	VGX_E_INVALID_ARG = 1,   /* null pointer, bad enum, count out of range */
	VGX_E_INVALID_PATH = 2,  /* command stream violates the path grammar (see vgx_pathset_create) */
	VGX_E_NONFINITE = 3,     /* NaN/Inf in path arguments (would hang the reference, path.cpp:109), or a draw whose
	                          * scale / tess_tol / fringe / stroke_width / mtx is NaN, Inf, negative, or scale, tess_tol <= 0,
	                          * or tess_tol / scale^2 < 1e-12 (checked on the device, reported in the status word) */
	VGX_E_NOSPACE = 4,       /* caller-provided output capacity too small; vgx_sizes holds the need */
	VGX_E_MESH_TOO_LARGE = 5,/* a mesh needs > 65536 vertices (uint16 indices, vg.cpp:734) */
	VGX_E_HIP = 6,           /* a HIP runtime call failed; vgx_last_hip_error() has the code */
	VGX_E_NO_DEVICE = 7,     /* no gfx950 device / HIP runtime unavailable */
	VGX_E_RANGE = 8,         /* batch exceeds 2^32-1 polyline vertices or commands; split the batch */
	VGX_E_INTERNAL = 9,      /* device-side protocol error (a wait inside the single-pass kernel timed out): a bug, report it */
	VGX_E_STALE = 10         /* vgx_tessellate: the draws no longer have the structure the last vgx_tessellate_count found (template
	                          * mode, see vgx_tessellate): a field other than mtx / colours / state_key of some draw differs from the
	                          * counted batch. Outputs are undefined; call vgx_tessellate_count on the new draws */
} vgx_status;

/* Path commands. One opcode per vg::pathXXX builder call (reference include/vg/path.h:24-35).
 * Arguments are float32, in the order of the reference's function parameters. */
typedef enum vgx_cmd {
	VGX_CMD_MOVE_TO = 0,   /* x, y                     pathMoveTo       path.cpp:62-78   */
	VGX_CMD_LINE_TO = 1,   /* x, y                     pathLineTo       path.cpp:80-84   */
	VGX_CMD_CUBIC_TO = 2,  /* c1x,c1y,c2x,c2y,x,y      pathCubicTo      path.cpp:86-182  */
	VGX_CMD_QUAD_TO = 3,   /* cx,cy,x,y                pathQuadraticTo  path.cpp:184-201 */
	VGX_CMD_CLOSE = 4,     /* -                        pathClose        path.cpp:707-726 */
	VGX_CMD_ARC_TO = 5,    /* x1,y1,x2,y2,r            pathArcTo        path.cpp:203-273 */
	VGX_CMD_ARC = 6,       /* cx,cy,r,a0,a1,dir(0|1)   pathArc          path.cpp:633-682 */
	VGX_CMD_RECT = 7,      /* x,y,w,h                  pathRect         path.cpp:275-286 */
	VGX_CMD_ROUNDED_RECT = 8,         /* x,y,w,h,r     pathRoundedRect  path.cpp:288-409 */
	VGX_CMD_ROUNDED_RECT_VARYING = 9, /* x,y,w,h,rtl,rtr,rbr,rbl        path.cpp:411-559 */
	VGX_CMD_CIRCLE = 10,   /* cx,cy,r                  pathCircle       path.cpp:561-597 */
	VGX_CMD_ELLIPSE = 11,  /* cx,cy,rx,ry              pathEllipse      path.cpp:599-631 */
	VGX_CMD_POLYLINE = 12, /* x0,y0,...,xn-1,yn-1      pathPolyline     path.cpp:684-705 */
	VGX_CMD_COUNT_ = 13
} vgx_cmd;

/* Values are the reference's vg::LineCap / vg::LineJoin (include/vg/vg.h:156-174); they are ABI. */
enum { VGX_CAP_BUTT = 0, VGX_CAP_ROUND = 1, VGX_CAP_SQUARE = 2 };
enum { VGX_JOIN_MITER = 0, VGX_JOIN_ROUND = 1, VGX_JOIN_BEVEL = 2 };

/* vgx_draw.fill_flags */
#define VGX_FILL_ENABLE 0x1u /* strokerConvexFill / strokerConvexFillAA per sub-path (vg.cpp:3099-3131) */
#define VGX_FILL_AA 0x2u
/* Index ORDER of strokerConvexFillAA meshes as the reference's default x86 build writes it (the SSE2 variant,
 * stroker.cpp:610-701: first fringe quad, then per fan triangle the triangle followed by the next edge's fringe quad, last
 * quad) instead of the scalar variant's (all fan triangles, then all fringe quads, stroker.cpp:769-795). Same triangles,
 * same counts, same vertices; for callers that compare index streams with an SSE build of the reference. Positions stay
 * the scalar build's (the SSE variant computes them with rcpps / rsqrtps approximations). */
#define VGX_FILL_INDEX_ORDER_SSE 0x100u
/* PathType::Concave fills (VG_FILL_FLAGS, include/vg/vg.h:229; ctxFillPath* src/vg.cpp:3133-3178): libtess2 triangulates them on
 * the CPU side of the caller, so a draw with VGX_FILL_CONCAVE and WITHOUT VGX_FILL_ENABLE produces no mesh in vgx_tessellate; its
 * mesh is built with vgx_flatten_* (contours) + libtess2 + vgx_concave_move / vgx_concave_emit and put at the draw's place in
 * the frame by vgx_merge. vgx_cmdlist_decode emits concave FillPath* commands as such draws (VGX_FILL_AA / VGX_FILL_EVEN_ODD say
 * which strokerConcaveFillEnd[AA] call and FillRule the reference would use). */
#define VGX_FILL_CONCAVE 0x10u
#define VGX_FILL_EVEN_ODD 0x20u
/* A user mesh (vg::indexedTriList, src/vg.cpp:4129-4175): the draw has no path and no GPU mesh; vgx_cmdlist_decode hands the mesh
 * itself over in vgx_cmdlist_out::tri_* (positions already through the state transform, as ctxIndexedTriList does with
 * batchTransformPositions) and vgx_merge puts it at the draw's place in the frame. */
#define VGX_FILL_TRILIST 0x40u
/* vgx_draw.stroke_flags */
#define VGX_STROKE_ENABLE 0x1u
#define VGX_STROKE_AA 0x2u
#define VGX_STROKE_THIN 0x4u /* strokerPolylineStrokeAAThin (vg.cpp:3417, 3464-3466); needs AA */
#define VGX_STROKE_CAP(flags) (((flags) >> 4) & 0x3u)
#define VGX_STROKE_JOIN(flags) (((flags) >> 6) & 0x3u)
#define VGX_STROKE_FLAGS(cap, join, aa, thin) \
	(VGX_STROKE_ENABLE | ((aa) ? VGX_STROKE_AA : 0u) | ((thin) ? VGX_STROKE_THIN : 0u) | ((uint32_t)(cap) << 4) | ((uint32_t)(join) << 6))

/* One path instance: what the reference does between vg::beginPath and vg::fillPath/strokePath for
 * one path under one state transform (vg.cpp:2969-2981 pathReset+strokerReset, vg.cpp:4957-4975
 * transformPath, vg.cpp:3061-3179 / 3401-3492 the stroker calls). 64 bytes, 16 dwords. */
typedef struct vgx_draw {
	uint32_t path;         /* index into the path set */
	uint32_t fill_flags;   /* VGX_FILL_* */
	uint32_t fill_color;   /* colour handed to strokerConvexFillAA */
	uint32_t stroke_flags; /* VGX_STROKE_* */
	uint32_t stroke_color; /* colour handed to strokerPolylineStrokeAA[Thin] */
	float stroke_width;    /* strokeWidth handed to strokerPolylineStroke[AA] (already scaled/clamped) */
	float scale;           /* pathReset/strokerReset scale (State::m_AvgScale) */
	float tess_tol;        /* tesselationTolerance (Context::m_TesselationTolerance) */
	float fringe;          /* fringeWidth (Context::m_FringeWidth) */
	float mtx[6];          /* 2x3 state transform [m0 m2 m4; m1 m3 m5] used by transformPath */
	uint32_t state_key;    /* draw-command assembly only (VGX_ASM_SPLIT_STATE): what allocDrawCommand / allocClipCommand compare
	                        * before merging a mesh into the previous command (vg.cpp:5376-5379, 5418-5428), folded by the host
	                        * into one word: DrawCommand::m_Type << 16 | m_HandleID in the low 20 bits, and above them a
	                        * generation the host bumps whenever it would set m_ForceNewDrawCommand / m_ForceNewClipCommand
	                        * (beginClip / endClip / resetClip / scissor changes, vg.cpp:3682-4026). 0 everywhere = one draw state */
} vgx_draw;

/* Path definitions ("path set"), host-side description handed to vgx_pathset_create.
 * cmd_arg_off has ncmd+1 entries: command k owns args[cmd_arg_off[k] .. cmd_arg_off[k+1]).
 * path p owns commands [path_cmd_begin[p], path_cmd_begin[p+1]). */
typedef struct vgx_pathset_desc {
	const uint8_t* cmd_type;        /* [ncmd]   vgx_cmd */
	const uint32_t* cmd_arg_off;    /* [ncmd+1] */
	const float* args;              /* [cmd_arg_off[ncmd]] */
	const uint32_t* path_cmd_begin; /* [npaths+1] */
	uint32_t npaths;
	uint32_t ncmd;
} vgx_pathset_desc;

/* vg::SubPath (include/vg/path.h:11-16) with batch-global addressing. 16 bytes. */
typedef struct vgx_subpath {
	uint64_t first_vertex; /* index into the batch's polyline vertex array */
	uint32_t num_vertices;
	uint32_t flags;        /* bit0 = isClosed */
} vgx_subpath;

/* Where one draw's data lives in the flatten output. 40 bytes. */
typedef struct vgx_draw_info {
	uint64_t first_poly_vertex;
	uint64_t first_subpath;
	uint64_t first_mesh;
	uint32_t num_poly_vertices; /* pathGetNumVertices */
	uint32_t num_subpaths;      /* pathGetNumSubPaths */
	uint32_t num_meshes;
	uint32_t flags;             /* bit0: went through the serial (exact, slow) lane path */
} vgx_draw_info;

/* vg::Mesh (include/vg/vg.h:353-360) with batch-global addressing. 32 bytes.
 * Meshes of a draw appear in the reference's call order: fill meshes by sub-path, then stroke
 * meshes by sub-path. Indices are mesh-local. */
typedef struct vgx_mesh {
	uint64_t first_vertex; /* into pos / color streams */
	uint64_t first_index;  /* into idx stream */
	uint32_t num_vertices;
	uint32_t num_indices;
	uint32_t draw;
	uint32_t subpath_kind; /* bits 0-27 sub-path index within the draw, bits 28-31 VGX_MESH_* */
} vgx_mesh;
enum { VGX_MESH_FILL = 0, VGX_MESH_FILL_AA = 1, VGX_MESH_STROKE = 2, VGX_MESH_STROKE_AA = 3, VGX_MESH_STROKE_AA_THIN = 4,
       VGX_MESH_CONCAVE_FILL_AA = 5 /* vgx_concave_emit */, VGX_MESH_TRILIST = 6 /* vgx_cmdlist_out::tri_meshes */ };

/* Totals of a batch. Filled by the *_count calls (host struct). */
typedef struct vgx_sizes {
	uint64_t num_poly_vertices;
	uint64_t num_subpaths;
	uint64_t num_meshes;
	uint64_t num_vertices;
	uint64_t num_indices;
	uint64_t num_serial_draws; /* draws that needed the exact serial lane path (degenerate input) */
	uint64_t num_cmd_instances;/* path commands summed over draws (flatten work items) */
	uint64_t num_elements;     /* polyline vertices summed over meshes (stroker work items) */
	uint64_t num_fill_elements;/* ... of which belong to convex-fill meshes (the rest to polyline strokes) */
	uint64_t num_drawcmds;     /* draw commands / vertex buffers of the assembly step (0 unless vgx_set_assembly armed it) */
} vgx_sizes;

/* Flatten output (pathGetVertices / pathGetSubPaths for every draw). NULL members are skipped. */
typedef struct vgx_flat_out {
	float* poly;              /* [cap_poly_vertices][2] */
	vgx_subpath* subpaths;    /* [cap_subpaths] */
	vgx_draw_info* draw_info; /* [ndraws] */
	uint64_t cap_poly_vertices;
	uint64_t cap_subpaths;
} vgx_flat_out;

/* Tessellation output: the three vg::Mesh streams concatenated mesh after mesh + a mesh table. */
typedef struct vgx_mesh_out {
	float* pos;       /* [cap_vertices][2] */
	uint32_t* color;  /* [cap_vertices]; non-AA meshes get the draw's colour on every vertex */
	uint16_t* idx;    /* [cap_indices] */
	vgx_mesh* meshes; /* [cap_meshes] */
	uint64_t cap_vertices;
	uint64_t cap_indices;
	uint64_t cap_meshes;
} vgx_mesh_out;

/* ---- draw-command assembly (optional next step of the frame) ------------------
 * What createDrawCommand_VertexColor / _Clip do after every stroker call (src/vg.cpp:5207-5244, 5297-5317): vertices go to
 * the current vertex buffer until it would exceed m_MaxVBVertices (allocVertices, :5321-5342), a new vertex buffer
 * forces a new draw command, meshes otherwise merge into the previous command when type and handle agree
 * (allocDrawCommand, :5359-5407), and indices are rebased by the vertices already in the COMMAND
 * (vgutil::batchTransformDrawIndices, vg_util.cpp:447-520). One vgx_drawcmd per draw command; 48 bytes. */
typedef struct vgx_drawcmd {
	uint64_t first_vertex;  /* where the command's vertices start in the pos / color / uv streams */
	uint64_t first_index;   /* DrawCommand::m_FirstIndexID: into the idx stream = the frame's single index buffer */
	uint64_t first_mesh;    /* first mesh merged into the command */
	uint32_t num_vertices;  /* DrawCommand::m_NumVertices */
	uint32_t num_indices;   /* DrawCommand::m_NumIndices */
	uint32_t num_meshes;
	uint32_t vertex_buffer; /* DrawCommand::m_VertexBufferID, counted from 0 for the batch */
	uint32_t first_vertex_in_vb; /* DrawCommand::m_FirstVertexID: offset inside its vertex buffer (0 for the buffer's first command) */
	uint32_t state_key;     /* the vgx_draw::state_key its meshes share (type / handle / generation); 0 without VGX_ASM_SPLIT_STATE */
} vgx_drawcmd;

enum { VGX_ASM_SPLIT_STATE = 1u }; /* vgx_assembly::flags: a change of vgx_draw::state_key between consecutive meshes starts a new
                                    * draw command inside the same vertex buffer (otherwise: one command per vertex buffer) */

typedef struct vgx_assembly {
	vgx_drawcmd* drawcmds;       /* DEVICE [cap_drawcmds]; 2 * vertices / max_vb_vertices + 2 (+ number of state changes) entries always suffice */
	uint64_t cap_drawcmds;
	uint64_t* dev_num_drawcmds;  /* DEVICE, may be NULL: receives the number of draw commands */
	uint32_t max_vb_vertices;    /* Config::m_MaxVBVertices (vg.cpp:726, <= 65536); 0 = 65536 */
	uint32_t flags;              /* VGX_ASM_* */
	/* the third vertex stream of createDrawCommand_VertexColor: every vertex gets the white-pixel UV (vg.cpp:5218-5225,
	 * vgutil::memset32 / memset64 of getWhitePixelUV): uv_bytes = 4 (VG_CONFIG_UV_INT16: int16 x 2) or 8 (float x 2) */
	void* uv;                    /* DEVICE [cap_vertices][uv_bytes], may be NULL */
	uint32_t uv_bytes;           /* 0 (no UV stream), 4 or 8 */
	uint32_t uv_value[2];        /* the constant, as raw bits (uv_value[1] unused for uv_bytes = 4) */
	uint32_t reserved;
} vgx_assembly;

typedef struct vgx_ctx vgx_ctx;         /* per-device context: scratch, scan storage, error state */
typedef struct vgx_pathset vgx_pathset; /* validated path definitions resident in device memory */

/* ---- context ------------------------------------------------------------------------------ */
/* Replaces createPath/createStroker (path.cpp:23-32, stroker.cpp:194-203): owns all scratch. */
int vgx_create(int device, vgx_ctx** out_ctx);
int vgx_destroy(vgx_ctx* ctx);
int vgx_last_hip_error(const vgx_ctx* ctx);
const char* vgx_status_string(int status);
uint32_t vgx_version(void);
/* Bytes of device scratch currently held by the context (polyline staging, tables, scan temp). */
uint64_t vgx_scratch_bytes(const vgx_ctx* ctx);

/* ---- path definitions --------------------------------------------------------------------- */
/* Validates and uploads a path set (round 6: on the DEVICE -- the four arrays go up as they are, the grammar checks are a
 * flagged reduction and every derived table is built by kernels, csrc/vgx_pathset.hip; what the reference does per command
 * while a path is recorded, path.cpp:62-84, 684-726, 761-784. The host reads one 32-byte record at the end; an invalid set
 * is handed to vgx_pathset_validate, which names the status). Grammar per path: first command must start a sub-path
 * (MOVE_TO, ARC, or a closed shape RECT, ROUNDED_RECT[_VARYING], CIRCLE, ELLIPSE); after CLOSE or a closed
 * shape the next command must start a sub-path again (the reference only VG_CHECKs this in debug
 * builds, path.cpp:82,88,764-765). Non-finite arguments are rejected. Synchronous.
 * A set whose paths are ALL made of MOVE_TO / LINE_TO / CLOSE only (polylines and polygons: pathMoveTo / pathLineTo /
 * pathClose, path.cpp:64-85, 707-726) also gets the polyline layout of every path here -- which commands add a vertex,
 * the vertex pathClose pops, the sub-path table: none of it depends on a draw -- and vgx_tessellate then moves such a
 * set's vertices through the draws' transforms without deciding anything again (csrc/vgx_thin.h; same output). */
int vgx_pathset_create(vgx_ctx* ctx, const vgx_pathset_desc* desc, vgx_pathset** out_ps);
/* Inspection (tests): one of the set's device tables copied to host memory; *bytes = its size (dst may be NULL to ask).
 * VGX_PS_TABLE_SCALARS: uint32[8] = longest path in commands, has serial paths, has empty paths, static thin layout, sub-paths,
 * npaths, ncmd, 0. */
enum { VGX_PS_TABLE_CMD_FLAGS = 0, VGX_PS_TABLE_SP_START = 1, VGX_PS_TABLE_PATH_FLAGS = 2, VGX_PS_TABLE_CMDREC = 3, VGX_PS_TABLE_PATH_SUB_BEGIN = 4,
       VGX_PS_TABLE_SUB_LAST_CMD = 5, VGX_PS_TABLE_CMDTHIN = 6, VGX_PS_TABLE_THIN_PATH = 7, VGX_PS_TABLE_THIN_SUB = 8, VGX_PS_TABLE_SCALARS = 9 };
int vgx_pathset_read_table(vgx_ctx* ctx, const vgx_pathset* ps, int which, void* dst, uint64_t cap_bytes, uint64_t* bytes);
/* The validation step of vgx_pathset_create alone (host only, needs no device). */
int vgx_pathset_validate(const vgx_pathset_desc* desc);
int vgx_pathset_destroy(vgx_ctx* ctx, vgx_pathset* ps);

/* ---- flatten: pathReset + path commands (+ optional transformPath) ------------------------- */
/* `draws` is a DEVICE pointer to ndraws vgx_draw records, 16-byte aligned (the kernels read a record as four 16-byte
 * words; hipMalloc memory and any offset that is a multiple of the 64-byte record are). apply_transform != 0 writes the
 * transformed polyline (what the stroker consumes); 0 writes pathGetVertices as-is.
 * _count runs count+scan and returns totals (synchronises the stream once to read them back);
 * _emit must follow with the same arguments and DEVICE output buffers of at least those sizes. */
int vgx_flatten_count(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, vgx_sizes* out_sizes, void* stream);
int vgx_flatten_emit(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, int apply_transform, const vgx_flat_out* out, void* stream);
/* Single asynchronous call, the form for steady state (like vgx_tessellate): the same ordered output as _count + _emit from ONE
 * walk over every cubic, no host round trip. `out` carries the caller's capacities (out->poly and out->subpaths must be given;
 * out->draw_info may be NULL); `dev_sizes` (DEVICE vgx_sizes, may be NULL) receives the totals -- num_poly_vertices,
 * num_subpaths, num_meshes, num_cmd_instances, num_serial_draws -- and `dev_status` (DEVICE uint32, may be NULL) VGX_OK /
 * VGX_E_NOSPACE (a capacity was too small: the totals say what is needed, the buffers' contents are undefined) / ... .
 * Replaces pathReset + the path commands + pathGetVertices / pathGetSubPaths (+ transformPath) for every draw, reference
 * src/path.cpp:44-78, 86-201, 684-726, src/vg.cpp:4957-4975. */
int vgx_flatten(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, int apply_transform, const vgx_flat_out* out,
                vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream);

/* ---- tessellate: flatten + transformPath + one strokerXXX call per sub-path per op --------- */
/* _count: flatten into context scratch, size every mesh, scan; returns totals (one stream sync).
 * _emit: writes pos/color/idx/meshes (DEVICE buffers). Must follow _count with the same batch. */
int vgx_tessellate_count(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, vgx_sizes* out_sizes, void* stream);
int vgx_tessellate_emit(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, const vgx_mesh_out* out, void* stream);
/* Single asynchronous call for steady state: count + scan + emit with NO host round trip. Output
 * capacities are checked on the device; `dev_sizes` (DEVICE vgx_sizes, may be NULL) receives the
 * totals and `dev_status` (DEVICE uint32, may be NULL) receives VGX_OK / VGX_E_NOSPACE / ... .
 * Template mode: when the last vgx_tessellate_count found that the draws repeat their first P draws (>= 32 times, > 2048 draws)
 * in everything but mtx, fill_color, stroke_color and state_key, it flattened the period ONCE in local space (the reference flattens before it transforms,
 * vg.cpp:4957-4975) and vgx_tessellate on this path set with any whole number of periods is one kernel: per instance the
 * template's vertices through the instance's transform, the stroker's per-element arithmetic, stores. Every call re-checks all
 * draw records against the counted period on the device; a draw that differs in another field ends the call with
 * VGX_E_STALE in dev_status (outputs undefined): count again. VGX_TMPL=0 in the environment at vgx_create turns the mode off.
 * The period may also come in a FEW flavours ("classes", at most 64: the same drawing at a handful of scales, say): every
 * instance then equals one class representative in the fields above, each class gets its own template, instances of different
 * classes have different sizes. Such a template belongs to the counted batch: vgx_tessellate takes it for the same number of
 * draws, with every instance still of the class it had at the count (else VGX_E_STALE). VGX_TMPL_CLASSES=0 turns this off.
 * Round joins (round 5): their arc points are counted on the TRANSFORMED polyline (stroker.cpp:1146, 1592), so a template batch with Round
 * joins has no fixed size -- every call counts them for the transforms it is given (two small kernels in front of the emit), dev_sizes holds
 * THIS call's totals, and a call whose output outgrows the caller's buffers ends with VGX_E_NOSPACE in dev_status (the need in dev_sizes,
 * nothing written) although the counted batch fitted. Templates of one class only; VGX_TMPL_ROUND=0 keeps such batches on the ordinary path. */
int vgx_tessellate(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream);

/* ---- stroker level: polylines in, meshes out ----------------------------------------------------- */
/* What the reference hands to strokerConvexFill[AA] / strokerPolylineStroke[AA|AAThin] (include/vg/stroker.h:29-72):
 * vertex lists that are ALREADY flattened and transformed. `poly` (DEVICE, [.][2]) holds the vertices, `subpaths`
 * (DEVICE) one record per vertex list {first_vertex, num_vertices, flags bit0 = isClosed}, `subpath_draw` (DEVICE)
 * the index of the vgx_draw whose fill_* / stroke_* / fringe / scale / tess_tol fields parameterise the calls for that
 * list (path and mtx are ignored). Mesh order: list after list, fill mesh (>= 3 vertices) before stroke mesh (>= 2).
 * vgx_mesh.subpath_kind carries the list index. _count sizes the output (one stream sync), _emit must follow. */
int vgx_stroke_count(vgx_ctx* ctx, const float* poly, const vgx_subpath* subpaths, const uint32_t* subpath_draw, uint64_t nsubpaths, const vgx_draw* draws, uint64_t ndraws, vgx_sizes* out_sizes, void* stream);
int vgx_stroke_emit(vgx_ctx* ctx, const float* poly, const vgx_subpath* subpaths, const uint32_t* subpath_draw, uint64_t nsubpaths, const vgx_draw* draws, uint64_t ndraws, const vgx_mesh_out* out, void* stream);

/* ---- draw-command assembly (see vgx_drawcmd / vgx_assembly above) ------------
 * Arms (asm_ != NULL) or disarms (NULL) assembly for the following vgx_tessellate_emit / vgx_tessellate calls on this
 * context. While armed, the uint16 indices in `idx` are vertex-buffer relative (mesh-local index + vertices in front of
 * the mesh inside its vertex buffer, uint16 wrap like the reference's cast, vg_util.cpp:447-520), `drawcmds` receives
 * one record per vertex buffer and vgx_sizes.num_drawcmds their number; pos / color / meshes are unchanged (the vertex
 * streams already are in vertex-buffer order). A mesh with more than max_vb_vertices vertices sets
 * VGX_E_MESH_TOO_LARGE (the reference VG_CHECKs it, vg.cpp:5323); a too small table VGX_E_NOSPACE. The struct is copied.
 * Template batches (see vgx_tessellate) are assembled too: the template pass writes the batch's mesh table for the partition
 * kernels and adds each mesh's base to the indices it emits. */
int vgx_set_assembly(vgx_ctx* ctx, const vgx_assembly* asm_);

/* ---- shape cache: tessellate a drawing once, submit it many times -------------
 * The reference keeps the meshes of a cached command list in the drawing's LOCAL space (addCachedCommand,
 * src/vg.cpp:5808-5841: positions times the inverse of the state transform at record time) and a later submission
 * only transforms them with the current state transform and appends positions, colours and indices to the frame
 * (submitCachedMesh, vg.cpp:6137-6166). All pointers below are DEVICE pointers. */
typedef struct vgx_cache_desc {   /* CommandListCache::m_Meshes as four streams: what vgx_tessellate[_emit] wrote */
	const float* pos;             /* [num_vertices][2], local space (after vgx_cache_localize) */
	const uint32_t* color;        /* [num_vertices] */
	const uint16_t* idx;          /* [num_indices], mesh-local */
	const vgx_mesh* meshes;       /* [num_meshes] */
	uint64_t num_meshes, num_vertices, num_indices;
} vgx_cache_desc;

typedef struct vgx_cache_instance { /* one submission of a cached command (clCacheRender, vg.cpp:5845-6135). 40 bytes */
	uint64_t first_mesh;          /* CachedCommand::m_FirstMeshID */
	uint32_t num_meshes;          /* CachedCommand::m_NumMeshes */
	uint32_t color;               /* the Color operand of the fill / stroke command being replayed (clCacheRender hands it to
	                               * submitCachedMesh, vg.cpp:5896-5902): meshes cached WITHOUT per-vertex colours -- the non-AA
	                               * flavours, numColors == 1 in addCachedCommand (:5826-5834) -- are drawn with it
	                               * (:6159-6160), the AA flavours with the colours stored at cache time */
	float mtx[6];                 /* State::m_TransformMtx at submission */
} vgx_cache_instance;

/* addCachedCommand: pos[v] <- inverse(draws[meshes[m].draw].mtx) * pos[v] for every vertex of every mesh, with the
 * reference's arithmetic (vgutil::invertMatrix3 in double precision, vg_util.cpp:14-33; transformPos2D). In place. */
int vgx_cache_localize(vgx_ctx* ctx, const vgx_draw* draws, uint64_t ndraws, float* pos, const vgx_mesh* meshes, uint64_t num_meshes, void* stream);
/* submitCachedMesh for `ninst` instances in order: for every mesh of every instance's range, positions through the
 * instance transform (batchTransformPositions), colours (stored ones for AA meshes, the instance's colour for non-AA meshes)
 * and indices copied; mesh records get the instance index as
 * `draw`. Asynchronous like vgx_tessellate (capacities checked on the device, totals in dev_sizes, status in
 * dev_status); honours vgx_set_assembly (createDrawCommand_VertexColor is what submitCachedMesh calls). */
int vgx_cache_submit(vgx_ctx* ctx, const vgx_cache_desc* cache, const vgx_cache_instance* instances, uint64_t ninst, const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream);

/* ---- concave fills with AA fringes -------------------------------------------------
 * strokerConcaveFillEndAA (src/stroker.cpp:868-1006) alternates libtess2 and the stroker's own loops:
 *   (1) tessTesselate(TESS_BOUNDARY_CONTOURS) of the contours added with strokerConcaveFillAddContour     [caller, CPU]
 *   (2) per boundary-contour vertex two fringe vertices + six indices; the contour vertex moves to the inner fringe
 *       vertex (:887-973)                                                                                  [vgx_concave_move / _emit]
 *   (3) tessAddContour of the moved contours, tessTesselate(TESS_POLYGONS)                                 [caller, CPU]
 *   (4) the interior appended behind the fringe, indices rebased (:976-994)                                [vgx_concave_emit]
 * libtess2 stays on the CPU side of the caller; these two calls do (2) and (4) for a BATCH of concave fills.
 * All pointers are DEVICE pointers. The boundary contours of one fill must be stored back to back in `contour_verts`
 * in tessGetElements order (what tessGetVertices returns), contours sorted by first_vertex. */
typedef struct vgx_contour {       /* one boundary contour of step (1): contourData[2i], contourData[2i+1]. 16 bytes */
	uint64_t first_vertex;         /* into contour_verts */
	uint32_t num_vertices;
	uint32_t fill;                 /* index of the concave fill it belongs to */
} vgx_contour;
typedef struct vgx_concave_fill {  /* one strokerConcaveFillBegin .. EndAA. 48 bytes */
	uint64_t first_contour;        /* its boundary contours [first_contour, first_contour + num_contours) */
	uint32_t num_contours;
	uint32_t color;                /* colour handed to strokerConcaveFillEndAA */
	float fringe;                  /* Stroker::m_FringeWidth */
	uint32_t num_tess_vertices;    /* step (3): tessGetVertexCount            (vgx_concave_emit only) */
	uint32_t num_tess_indices;     /*           tessGetElementCount * 3 */
	uint32_t reserved;
	uint64_t first_tess_vertex;    /* into tess_pos */
	uint64_t first_tess_index;     /* into tess_idx */
} vgx_concave_fill;
/* Step (2), first half: moved[v] = the inner fringe vertex of contour vertex v (what the reference writes back into the
 * contour before it hands it to libtess2 again). `moved` has the layout of contour_verts. Asynchronous. */
int vgx_concave_move(vgx_ctx* ctx, const float* contour_verts, uint64_t num_contour_vertices, const vgx_contour* contours, uint64_t ncontours,
                     const vgx_concave_fill* fills, uint64_t nfills, float* moved, void* stream);
/* Steps (2) + (4): one mesh per fill = [2 vertices, 6 indices per contour vertex][interior from tess_pos / tess_idx with
 * indices rebased by the fringe's vertex count], meshes concatenated in fill order; mesh records carry draw = fill index,
 * kind VGX_MESH_CONCAVE_FILL_AA. contour_verts are the ORIGINAL boundary contours (not the moved ones). Asynchronous
 * like vgx_tessellate (capacities checked on the device, totals in dev_sizes, status in dev_status). */
int vgx_concave_emit(vgx_ctx* ctx, const float* contour_verts, uint64_t num_contour_vertices, const vgx_contour* contours, uint64_t ncontours,
                     const vgx_concave_fill* fills, uint64_t nfills, const float* tess_pos, const uint16_t* tess_idx,
                     const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream);

/* ---- merging external meshes into a frame -----------------------------------------------------------------
 * The reference appends every mesh to the frame in submission order (createDrawCommand_VertexColor, src/vg.cpp:5207-5244).
 * vgx_merge builds that order from two mesh sequences that are each sorted by draw: `a` = what vgx_tessellate[_emit] wrote for
 * the frame's draws, `b` = meshes built elsewhere for draws that have none in `a` (concave fills: vgx_concave_emit), with
 * b_draw[j] (DEVICE, may be NULL: b->meshes[j].draw) = the frame draw of b's mesh j. Output = both sequences interleaved by draw
 * index (a mesh of `a` before a mesh of `b` of the same draw), streams copied, mesh records renumbered (first_vertex /
 * first_index = the merged offsets, draw = the frame draw). Honours vgx_set_assembly like vgx_tessellate does (`draws` / ndraws:
 * the frame's draw records, DEVICE, only read for VGX_ASM_SPLIT_STATE; may be NULL otherwise). All pointers are DEVICE pointers;
 * the num_* members of `a` / `b` are host values. Asynchronous (capacities checked on the device, totals in dev_sizes, status in
 * dev_status); a sequence that is not sorted by draw sets VGX_E_INVALID_ARG. */
int vgx_merge(vgx_ctx* ctx, const vgx_cache_desc* a, const vgx_cache_desc* b, const uint32_t* b_draw, const vgx_draw* draws, uint64_t ndraws,
              const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream);
/* The same with per-vertex UVs for the meshes of `b` (user meshes with texture coordinates): b_uv (DEVICE, [b->num_vertices] x the
 * armed assembly's uv_bytes, or NULL) overwrites the white-pixel UV the assembly step writes for every vertex. Needs an armed
 * assembly with a UV stream to have an effect. */
int vgx_merge_uv(vgx_ctx* ctx, const vgx_cache_desc* a, const vgx_cache_desc* b, const uint32_t* b_draw, const void* b_uv, const vgx_draw* draws, uint64_t ndraws,
                 const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream);

/* ---- command-list byte-code as input ---------------------------------------------
 * vg::CommandList::m_CommandBuffer as the reference's cl* functions write it (src/vg.cpp:243-247, 2403-2690, 5694-5723):
 * {CommandHeader{uint32 type, uint32 size}, 16-byte aligned}{payload, 16-byte aligned}... in HOST memory.
 * vgx_cmdlist_decode replays it the way ctxSubmitCommandList does (:4273-4637) into what the batch entry points take: one
 * path per BeginPath group (vgx_pathset_desc arrays) and one vgx_draw per fill / stroke command -- all six of them:
 * FillPathColor / Gradient / ImagePattern (ctxFillPath*, :3061-3399), StrokePathColor / Gradient / ImagePattern
 * (ctxStrokePath*, :3401-3668) -- with the interpreter's state folded in: transform stack (PushState / PopState /
 * Transform* / SetViewBox, :3934-4122; the transform is latched at the path's first fill / stroke like transformPath
 * :4957-4975), global alpha, stroke width scaling / clamping / Thin switch, scissor (per draw, vgx_draw_state), clip
 * regions (BeginClip / EndClip / ResetClip, :3670-3709: the draws recorded inside are non-AA black DrawCommand::Type::Clip
 * draws), gradients and image patterns created by the list (vgx_paint records, :3711-3932), command culling
 * (CommandListFlags::AllowCommandCulling), and nested lists (SubmitCommandList, :4611-4620, through `lists`).
 * vgx_draw::state_key = generation << 20 | DrawCommand::Type << 16 | handle: what allocDrawCommand / allocClipCommand
 * compare before merging (:5359-5460); the generation changes whenever the reference sets m_ForceNewDrawCommand /
 * m_ForceNewClipCommand between two draws (scissor changes, PopState onto a different scissor, EndClip, ResetClip).
 * Concave fills (PathType::Concave) become draws with VGX_FILL_CONCAVE [| VGX_FILL_AA] [| VGX_FILL_EVEN_ODD] and no
 * VGX_FILL_ENABLE: vgx_tessellate makes no mesh for them, the caller builds it (vgx_flatten_* -> libtess2 -> vgx_concave_move /
 * vgx_concave_emit) and vgx_merge puts it at the draw's place in the frame.
 * IndexedTriList commands become draws with VGX_FILL_TRILIST; their meshes come back in vgx_cmdlist_out::tri_*.
 * Commands without an equivalent here are counted in num_skipped and otherwise ignored: Text / TextBox,
 * path commands issued after a path's first fill / stroke
 * without a new BeginPath (the reference VG_CHECKs this, :2984-3059), nested lists without a table entry.
 * Host only, no device needed; re-entrant (no shared state between calls). `bytes` must be 4-byte aligned (the reference's
 * buffers are 16-byte aligned). Call with the array members NULL to get the counts, allocate, call again. */
typedef struct vgx_cmdlist_ref {   /* one vg::CommandList, addressed by CommandListHandle::idx (SubmitCommandList) */
	const void* bytes;             /* HOST CommandList::m_CommandBuffer */
	uint32_t size;                 /* m_CommandBufferPos */
	uint32_t flags;                /* m_Flags (VGX_CL_*) */
} vgx_cmdlist_ref;
enum { VGX_CL_CACHEABLE = 1u,      /* CommandListFlags::Cacheable: fills / strokes ignore the global alpha and transparent
                                    * colours are not dropped while the list populates its cache (hasCache, :3063-3075) */
       VGX_CL_ALLOW_CULLING = 2u, /* CommandListFlags::AllowCommandCulling (:4299-4300, 4548-4577) */
       VGX_CL_UV_FLOAT = 0x200u,  /* not a reference flag: the build's uv_t is float (VG_CONFIG_UV_INT16 = 0, include/vg/vg.h:27-29): IndexedTriList payloads
                                    * carry 8 bytes of UV per vertex instead of 4 */
       VGX_CL_SCISSOR_SET = 0x100u };/* not a reference flag: vgx_cmdlist_state::scissor holds a rectangle even when it is all zero (a real
                                    * empty scissor left by an earlier list of the frame); set it when chaining vgx_cmdlist_out::end_scissor */
typedef struct vgx_cmdlist_state { /* the Context / State values at submission */
	float mtx[6];          /* State::m_TransformMtx */
	float global_alpha;    /* State::m_GlobalAlpha */
	float tess_tol;        /* Context::m_TesselationTolerance */
	float fringe;          /* Context::m_FringeWidth */
	float canvas_width;    /* Context::m_CanvasWidth / Height (SetViewBox, scissor clamps) */
	float canvas_height;
	uint32_t flags;        /* VGX_CL_* of the list being decoded */
	float scissor[4];      /* State::m_ScissorRect; all zero WITHOUT VGX_CL_SCISSOR_SET in flags = {0, 0, canvas_width, canvas_height} (resetScissor) */
	uint32_t first_gradient;      /* Context::m_NextGradientID at submission (local handles are relative to it) */
	uint32_t first_image_pattern; /* Context::m_NextImagePatternID */
	uint32_t max_gradients;       /* Config::m_MaxGradients, 0 = 64 */
	uint32_t max_image_patterns;  /* Config::m_MaxImagePatterns, 0 = 64 */
	uint32_t max_depth;           /* Config::m_MaxCommandListDepth, 0 = 16 */
	uint32_t num_lists;           /* entries of `lists` */
	const vgx_cmdlist_ref* lists; /* HOST handle -> list table for SubmitCommandList; NULL: nested lists are skipped */
	uint16_t prev_cmd_scissor[4]; /* scissor of the frame's last draw command before this list (PopState rule, :3950-3965) */
	uint32_t prev_cmd_valid;      /* 0: the frame has no draw command yet */
	uint32_t first_generation;    /* generation of the first draw's state_key (chain successive decodes of one frame) */
	/* Context::m_ClipState / m_RecordClipCommands at submission: a clip region outlives the list that recorded it (vg.cpp:71-76,
	 * 3670-3709). All zero = no region. Indices are in the frame's draw numbering: this decode's draw i is draw draw_base + i. */
	uint32_t clip_valid;          /* 1: a region is active (m_ClipState.m_FirstCmdID != ~0) */
	uint32_t clip_rule;
	uint32_t clip_first_draw;
	uint32_t clip_num_draws;
	uint32_t clip_recording;      /* 1: submitted between BeginClip and EndClip */
	uint32_t draw_base;           /* draws decoded earlier in this frame */
	uint32_t white_uv[2];         /* getWhitePixelUV as raw words (one word with int16 UVs): the UV of IndexedTriList vertices that come without UVs (vg.cpp:4148-4156) */
	uint32_t font_image;          /* Context::m_FontImages[0].idx: the image an IndexedTriList with an invalid handle is drawn with (:4131-4133) */
} vgx_cmdlist_state;
typedef struct vgx_draw_state {   /* per draw: what allocDrawCommand copies into the DrawCommand (vg.cpp:5391-5400). 24 bytes */
	uint16_t scissor[4];          /* (uint16_t) State::m_ScissorRect */
	uint32_t clip_rule;           /* ClipState::m_Rule (0 In, 1 Out) */
	uint32_t clip_first_draw;     /* the active clip region = the draws of type Clip among [clip_first_draw, + clip_num_draws) of
	                               * this decode (the reference stores the range of clip COMMANDS they merge into; gradient and
	                               * image-pattern paints inside BeginClip .. EndClip are ordinary draws and may lie between
	                               * them); 0xFFFFFFFF = none. A draw recorded while the region is still open sees it empty
	                               * (clip_num_draws = 0), as in the reference (vg.cpp:3670-3697) */
	uint32_t clip_num_draws;
	uint32_t raw_color;           /* the Color operand of the fill / stroke command as recorded (0 for gradient paints): what a
	                               * replay from the shape cache uses for non-AA meshes (vgx_cache_instance::color) */
} vgx_draw_state;
typedef struct vgx_paint {        /* vg::Gradient / vg::ImagePattern as the Create* calls compute them (vg.cpp:84-96). 96 bytes */
	uint32_t type;                /* DrawCommand::Type of the draws that use it: 1 ColorGradient, 2 ImagePattern */
	uint32_t handle;              /* the id in the low 16 bits of those draws' state_key */
	float matrix[9];              /* m_Matrix */
	float params[4];              /* Gradient::m_Params {extent.x, extent.y, radius, feather} */
	float inner_color[4];
	float outer_color[4];
	uint32_t image;               /* ImagePattern::m_ImageHandle */
} vgx_paint;
typedef struct vgx_cmdlist_out {
	uint8_t* cmd_type;        /* HOST [cap_cmds]      -> vgx_pathset_desc.cmd_type */
	uint32_t* cmd_arg_off;    /* HOST [cap_cmds + 1] */
	float* args;              /* HOST [cap_args] */
	uint32_t* path_cmd_begin; /* HOST [cap_paths + 1] */
	vgx_draw* draws;          /* HOST [cap_draws]; vgx_draw.path indexes the paths produced here */
	vgx_draw_state* draw_state; /* HOST [cap_draws], may be NULL */
	vgx_paint* paints;        /* HOST [cap_paints], may be NULL */
	uint32_t cap_cmds, cap_args, cap_paths, cap_draws, cap_paints;
	uint32_t num_cmds, num_args, num_paths, num_draws, num_paints; /* out */
	uint32_t num_skipped;     /* out: commands without an equivalent in this path */
	uint32_t next_gradient;   /* out: Context::m_NextGradientID / m_NextImagePatternID after the list */
	uint32_t next_image_pattern;
	uint32_t next_generation; /* out: first_generation for the next decode of the same frame */
	float end_mtx[6];         /* out: State::m_TransformMtx / m_GlobalAlpha after the list (state changes of a list leak into
	                           * its caller unless VG_CONFIG_COMMAND_LIST_PRESERVE_STATE, vg.cpp:4323-4325) */
	float end_global_alpha;
	uint32_t end_clip_valid;  /* out: the clip state after the list, for the next decode's vgx_cmdlist_state::clip_* */
	uint32_t end_clip_rule, end_clip_first_draw, end_clip_num_draws, end_clip_recording;
	float end_scissor[4];     /* out: State::m_ScissorRect after the list (chain it with VGX_CL_SCISSOR_SET: it may be a real empty rectangle) */
	uint32_t reserved;
	/* IndexedTriList commands (user meshes, vg.cpp:4129-4175 / 4461-4477): one draw each (VGX_FILL_TRILIST, state_key = Textured |
	 * image) and the mesh in these HOST arrays, ready to be uploaded as a sequence for vgx_merge: positions through the state
	 * transform at the command (batchTransformPositions), one colour per vertex (a single colour replicated, :4160-4165), the
	 * command's UVs or the white-pixel UV, indices mesh-local; tri_meshes[k].draw = the draw's index in this decode,
	 * subpath_kind = VGX_MESH_TRILIST << 28. The count pass reports num_tri_*; a store pass over a list that holds such commands
	 * without these arrays (or with too small ones) returns VGX_E_NOSPACE -- never a frame with a mesh silently missing.
	 * tri_uv alone may be NULL (no UVs wanted). */
	float* tri_pos;           /* HOST [cap_tri_vertices][2] */
	uint32_t* tri_color;      /* HOST [cap_tri_vertices] */
	void* tri_uv;             /* HOST [cap_tri_vertices][4 or 8 bytes (VGX_CL_UV_FLOAT)] */
	uint16_t* tri_idx;        /* HOST [cap_tri_indices] */
	vgx_mesh* tri_meshes;     /* HOST [cap_tri_meshes] */
	uint32_t cap_tri_vertices, cap_tri_indices, cap_tri_meshes;
	uint32_t num_tri_vertices, num_tri_indices, num_tri_meshes; /* out */
} vgx_cmdlist_out;
int vgx_cmdlist_decode(const void* bytes, uint32_t size, const vgx_cmdlist_state* state, vgx_cmdlist_out* out);

/* Diagnostics of the last asynchronous call on this context: the device status word and, when a kernel of
 * vgx_tessellate gave up, why (reason = one of the VGX_FAIL_* codes of csrc/vgx_internal_types.h: a table of
 * the kernel was too small for the batch -- run vgx_tessellate_count on a batch like it --, the polyline heap or the
 * caller's output capacity was exhausted, ...). Synchronises `stream`. Not needed on the happy path. */
typedef struct vgx_failure_info {
	uint32_t status;        /* vgx_status of the device status word */
	uint32_t reason;        /* 0 = none */
	uint32_t aux;           /* reason specific (a count) */
	uint32_t segment_items; /* (historic name) flatten kernel the last vgx_tessellate_count chose for batches like its own:
	                         * 0 = k_flatten_build (one lane per path command), 1 = k_flatten_inst, periodic draws, 2 = k_flatten_inst, draws
	                         * sorted by path, 3 = k_flatten_inst, draws sorted by (path, tolerance class) -- instances of different scales,
	                         * 4 = k_flatten_inst, periodic draws with the INSTANCES sorted by tolerance class, 5 = template mode (no flatten
	                         * per call), 6 = k_flatten_thin: like 0 for a path set of moveTo / lineTo / close paths only, whose polyline
	                         * layout was decided when the set was created */
	uint64_t segment;       /* work item (segment / task) that failed first */
	uint64_t prof[16];      /* -DVGX_INST_PROFILE builds of libvgx only (else 0): wave clock ticks (100 MHz) summed over all waves
	                         * per phase of k_flatten_inst (profiles/inst_phases.py) */
} vgx_failure_info;
int vgx_get_failure_info(vgx_ctx* ctx, vgx_failure_info* out, void* stream);

/* ---- multi-GPU: partition of a batch into contiguous draw ranges of about equal predicted output ----------------
 * "Partitioning: contiguous ranges of path instances per GPU ...; for heterogeneous batches balance on the count-pass result
 * (predicted out-verts)". Runs the flatten count pass over the whole batch (per-draw polyline vertex counts; no output buffers,
 * no mesh scratch), weights every draw with polyline vertices x output vertices per polyline vertex of its fill / stroke
 * flavour (+ 1), and cuts the draw sequence where the weight prefix crosses k / nparts of the total:
 *   out_bounds[0] = 0 <= out_bounds[1] <= ... <= out_bounds[nparts] = ndraws   (HOST, nparts + 1 entries)
 *   out_weights[k] = predicted weight of part k (HOST, nparts entries; may be NULL)
 * Rank r then tessellates draws [out_bounds[r], out_bounds[r + 1]); rank order = draw order, so the gathered streams are the
 * single-GPU result. Homogeneous batches (Tiger x K) come out as equal instance counts; when the draws repeat one sequence of
 * paths (a drawing submitted for many instances, at whatever scales) every cut falls BETWEEN instances, so that each part is
 * again a batch of whole instances for the instanced / template paths of vgx_tessellate. Synchronises the stream.
 * vgx_partition is a count call over the WHOLE batch: it replaces what an earlier vgx_tessellate_count left in the context
 * (scratch sizes, the instanced / template classification). Every rank calls vgx_tessellate_count on its own range afterwards. */
int vgx_partition(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, uint32_t nparts, uint64_t* out_bounds, uint64_t* out_weights, void* stream);

/* ---- multi-GPU: gather of the per-rank streams to one root over RCCL / xGMI ---------------------------------
 * Independent path instances shard embarrassingly: one process per GPU, rank r tessellates a contiguous range of the
 * draws, no data-path collective. Indices are mesh-local, so the single-GPU result is the per-rank streams concatenated
 * in rank order; only the mesh table's first_vertex / first_index / draw need the rank's base added. This is the one
 * exchange step, for a C / C++ host that owns an RCCL communicator (`rccl_comm` is its ncclComm_t, created on the
 * context's device; librccl is bound at the first call from the library already loaded in the process, libvgx.so itself
 * has no link-time dependency on it). vg-renderer_amd/dist.py is the same layout over torch.distributed.
 *
 *   vgx_gather_sizes  all-gathers the four per-rank totals on the device (ncclAllGather of 4 x uint64 through context
 *                     scratch) and copies them to `all` (HOST, [nranks], rank order). Synchronises `stream`. Batches that
 *                     keep their shape from frame to frame call it once.
 *   vgx_gather        enqueues on `stream`, without any host synchronisation: on every other rank four ncclSend (positions,
 *                     colours, indices, mesh table) to `root`; on the root the matching ncclRecv straight into `global` at
 *                     each rank's offset, all inside ONE group (each peer -> root transfer rides its own xGMI link), a
 *                     device-to-device copy of the root's own block, and the rebase of the gathered mesh table (one
 *                     kernel). `global` is only read on the root (capacities checked against the totals: VGX_E_NOSPACE).
 *                     To overlap the gather of frame i with the tessellation of frame i + 1, call it on a second stream
 *                     with double-buffered outputs: nothing in it touches context scratch that vgx_tessellate uses.
 *                     Transfers go out in pieces of at most VGX_GATHER_CHUNK_MB (default 256 MiB), one group per piece. The
 *                     piece size is part of the wire protocol (sender and root cut a stream the same way) and is read once
 *                     per process from the environment: it must be the same on every rank; vgx_gather_sizes compares the
 *                     ranks' values and returns VGX_E_INVALID_ARG when they differ.
 * Errors: VGX_E_NO_DEVICE when no RCCL library can be bound, VGX_E_HIP when an RCCL call fails (vgx_last_hip_error() then
 * holds 10000 + the ncclResult_t). */
typedef struct vgx_rank_sizes { uint64_t num_vertices, num_indices, num_meshes, num_draws; } vgx_rank_sizes;
int vgx_gather_sizes(vgx_ctx* ctx, void* rccl_comm, const vgx_rank_sizes* mine, vgx_rank_sizes* all, void* stream);
int vgx_gather(vgx_ctx* ctx, void* rccl_comm, int root, const vgx_mesh_out* local, const vgx_rank_sizes* all, const vgx_mesh_out* global, void* stream);
/* vgx_gather with explicit placement: rank r's block (all[r] elements of `local` on rank r) lands at place[r] in `global` --
 * {num_vertices, num_indices, num_meshes} = first vertex / index / mesh of the block, num_draws = what is added to the `draw`
 * of its mesh records (vgx_gather = the exclusive prefix of `all`). This is what a frame tessellated in TILES needs: a rank cuts
 * its draws into sub-batches, tessellates tile t into the local buffers behind tile t - 1 and gathers tile t (local = a view of
 * the tile's part of the buffers, place = the rank's base + the tiles in front) on a second stream while tile t + 1 is being
 * tessellated -- the gathered frame is the same bytes, in the same order. Messages larger than VGX_GATHER_CHUNK_MB (default
 * 256 MiB) are split into pieces, one RCCL group per piece index. Capacities are checked against place + all. */
int vgx_gather_at(vgx_ctx* ctx, void* rccl_comm, int root, const vgx_mesh_out* local, const vgx_rank_sizes* all, const vgx_rank_sizes* place,
                  const vgx_mesh_out* global, void* stream);

/* Static batches (default off; also VGX_TMPL_BATCH=1). The caller promises that the batches it submits between two vgx_tessellate_count
 * calls keep their STRUCTURE -- the same paths with the same fill / stroke styles, widths, scale, tolerance and fringe at the same
 * positions of the draw list -- and only move transforms, colours and state keys: the draw list a retained scene produces frame after
 * frame (vg::submitCommandList of an unchanged list under a new camera; an instanced scene after culling, whose draws no longer repeat
 * a period). vgx_tessellate_count then flattens the whole draw list ONCE, in local space (the reference flattens before transformPath,
 * src/vg.cpp:4957-4975), and keeps it as one template (vgx_tmpl.hip: local polyline, mesh and element tables: ~30 bytes per output
 * vertex of device memory); vgx_tessellate is then ONE kernel per call -- every draw record verified against the counted one,
 * transformPos2D, the stroker -- instead of flatten + scans + fill + stroke. A structural change ends the call with VGX_E_STALE
 * (nothing usable in the buffers): count again. Batches above 2^29 vertices or 2^31 indices / elements keep the ordinary pipeline.
 * Results are the same bytes either way. */
int vgx_set_static_batches(vgx_ctx* ctx, int enable);

/* Per-kernel timing of the last vgx_tessellate.. / vgx_flatten.. sequence, measured with HIP events
 * on the stream the kernels ran on. Enable before the call; read after synchronising. */
#define VGX_MAX_STAGES 16
typedef struct vgx_stage_times {
	uint32_t num_stages;
	float ms[VGX_MAX_STAGES];
	const char* name[VGX_MAX_STAGES];
} vgx_stage_times;
int vgx_set_profiling(vgx_ctx* ctx, int enable);
int vgx_get_stage_times(vgx_ctx* ctx, vgx_stage_times* out);
/* The same, averaged over the last `ncalls` profiled calls of this context (at most VGX_PROF_RING, and only calls with the
 * same stage sequence as the last one): a caller that times K back-to-back calls reads the per-kernel durations of those
 * very calls afterwards, without having synchronised between them. Read after synchronising. */
#define VGX_PROF_RING 32
int vgx_get_stage_times_avg(vgx_ctx* ctx, vgx_stage_times* out, uint32_t ncalls);

#ifdef __cplusplus
}
#endif
#endif /* VGX_H */
