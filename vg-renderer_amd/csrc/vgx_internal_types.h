// vgx_internal_types.h -- plain data layouts shared by device code and the host unit-test build.
#ifndef VGX_INTERNAL_TYPES_H
#define VGX_INTERNAL_TYPES_H

#include <stdint.h>
#include "../../include/vgx.h"

// ---- path set resident in HBM ----------------------------------------------------------------
// Same SoA arrays as vgx_pathset_desc plus two per-command arrays derived on the host when the set is
// created (static structure of the command stream, independent of draws):
//   cmd_flags    bit0 STARTS_SUB   command opens a sub-path (MOVE_TO, closed shapes, leading ARC)
//                bit1 LAST_IN_SUB  next command opens a sub-path or the path ends
//                bit2 NEXT_IS_CLOSE
//                bit3 LAST_IN_PATH
//   cmd_sp_start absolute index of the command that opened this command's sub-path
//   path_flags   bit0 SERIAL       path contains ARC / ARC_TO: their end points are computed, so the
//                                  lane-parallel start-point gather is impossible -> exact serial lane path
#define VGX_CF_STARTS_SUB 0x1u
#define VGX_CF_LAST_IN_SUB 0x2u
#define VGX_CF_NEXT_IS_CLOSE 0x4u
#define VGX_CF_LAST_IN_PATH 0x8u
#define VGX_PF_SERIAL 0x1u
#define VGX_PF_THIN 0x2u   // every command of the path is MOVE_TO / LINE_TO / CLOSE: its lanes read the 16-byte thin records

// Thin record, one per command (used for VGX_PF_THIN paths only): a polyline drawn with lineTo calls -- 10 000 paths x 1 000
// segments -- read a 64-byte VgxCmdRec per 8-byte output vertex (FETCH 4-8x the command bytes, round 4). 16 bytes:
//   meta  type | flags << 8
//   x, y  MOVE_TO / LINE_TO: the command's point (= the NEXT command's start point);  CLOSE: the first point of its sub-path
// so a lane finds its start point in the record in front of it and pathClose's first point in the record behind it.
struct VgxCmdThin
{
	uint32_t meta;
	float x, y;
	uint32_t pad;
};

// One fixed-size record per path command, built on the host at upload: everything a lane needs for its command in
// ONE 64-byte load (the SoA arrays above stay for the serial path and for POLYLINE's variable arguments).
struct VgxCmdRec
{
	uint32_t type;      // vgx_cmd
	uint32_t flags;     // VGX_CF_*
	uint32_t na;        // argument count
	uint32_t arg_off;   // into args (POLYLINE)
	float start[2];     // previous command's end point (the lane's start point; unused by sub-path starters)
	float a[8];         // arguments 0..7; for CLOSE a[6..7] = first point of its sub-path (the MOVE_TO point)
	float pad[2];
};

struct VgxPathSetDev
{
	const VgxCmdRec* cmdrec;
	const VgxCmdThin* cmdthin;      // [ncmd + 2] (one padding record in front: command 0 reads its predecessor's)
	const struct VgxThinPath* thin_path; // [npaths] static polyline layout of MOVE_TO / LINE_TO / CLOSE paths (vgx_thin.h), filled when EVERY path of the set is one
	const struct VgxThinSub* thin_sub;   // [sub-paths of the set] indexed like sub_last_cmd
	const uint8_t* cmd_type;
	const uint8_t* cmd_flags;
	const uint32_t* cmd_arg_off;
	const uint32_t* cmd_sp_start;
	const float* args;
	const uint32_t* path_cmd_begin;
	const uint8_t* path_flags;
	const uint32_t* path_sub_begin; // [npaths + 1] first entry of the path in sub_last_cmd
	const uint32_t* sub_last_cmd;   // command index (relative to the path's first command) of every LAST_IN_SUB command
	uint32_t npaths;
	uint32_t ncmd;
};

// ---- per-command-instance word written by the count pass, read by the emit pass ------------------
// bits 0..27 vertex count contributed by the command; bit28 EXISTS (command opened a sub-path that
// really exists), bit29 CLOSED (this command closed its sub-path), bit30 POP (CLOSE removed the
// previous vertex, pathClose path.cpp:716-725).
#define VGX_CC_COUNT_MASK 0x0FFFFFFFu
#define VGX_CC_EXISTS 0x10000000u
#define VGX_CC_CLOSED 0x20000000u
#define VGX_CC_POP 0x40000000u

// ---- one mesh to generate (sub-path x op), written by flatten-emit, consumed by the stroker ---------
struct VgxMeshDesc
{
	uint64_t poly_first; // first polyline vertex (batch-global)
	uint32_t poly_n;
	uint32_t draw;
	uint32_t subpath;    // sub-path index within the draw
	uint32_t kind;       // bits 0-7 VGX_MESH_*, bit 8 closed, bits 9-10 effective cap, bits 11-12 effective join, bit 13 SSE index order (FILL_AA)
};
#define VGX_MD_SSE_ORDER(k) (((k) >> 13) & 1u)
#define VGX_MD_KIND(k) ((k) & 0xFFu)
#define VGX_MD_CLOSED(k) (((k) >> 8) & 1u)
#define VGX_MD_CAP(k) (((k) >> 9) & 3u)
#define VGX_MD_JOIN(k) (((k) >> 11) & 3u)

// Per-mesh constants computed once by k_mesh_prepare so that the element kernels do not touch the draw record:
//   fills   f0 = aa = fringe/2 * sign(first triangle) (stroker.cpp:721-723)
//   strokes f0 = hsw, f1 = hsw_aa, f2 = fringe        (stroker.cpp:1011-1012, 1396-1397, 1999)
struct VgxMeshPrep
{
	float f0, f1, f2;
	uint32_t color;
};

#define VGX_LONG_STROKE 128u
#define VGX_MESH_NEEDS_COUNT 0xFFFFFFFFu // mtab.num_vertices marker: Round joins, sized by k_round_sizes

// Sub-path record of the single-pass flatten, stored sparsely at the command-instance index of the sub-path's last
// command: one 16-byte record = one sector for k_flatten_gather to fetch (two separate arrays cost two).
struct VgxSubRec
{
	uint64_t first; // heap index of the sub-path's first polyline vertex
	uint32_t info;  // vertex count | closed << 31
	uint32_t pad;   // VGX_ORIENT_*: sign of the sub-path's first triangle (fill orientation, stroker.cpp:721-723), when the writer knew it
};
#define VGX_ORIENT_KNOWN 1u
#define VGX_ORIENT_POS 2u
#define VGX_ORIENT_NEG 4u

#define VGX_INST_POOLS 16
// ---- batch totals kept in device memory (mirrors vgx_sizes + internal counters) -------------------
struct VgxTotals
{
	vgx_sizes sizes;
	uint32_t status;       // vgx_status, sticky (first error wins)
	uint32_t num_round_meshes; // meshes with Round joins (their sizes need the geometry)
	unsigned long long poly_heap_cursor; // BUILD mode: bump allocator of the polyline heap (vertices)
	unsigned long long long_subpath_vertices; // count pass: vertices in sub-paths longer than VGX_LONG_SUBPATH (heap sizing)
	unsigned long long num_serial_list;  // BUILD mode: entries of serial_list (draws k_flatten_serial has to redo)
	// instanced batches (vgx_inst.hip)
	unsigned long long inst_long_subpath_vertices; // count pass: vertices in sub-paths longer than VGX_INST_LONG_SUBPATH (heap sizing)
	unsigned long long inst_detect_inv;  // vgx_tessellate_count: ~0 - (index of the first repetition of draws[0].path); 0 = none
	unsigned long long inst_ticket[VGX_INST_POOLS * 16]; // vgx_tessellate: next task of every task pool of k_flatten_inst, one counter per 128 bytes
	unsigned long long inst_num_tasks;   // grouped mode (k_inst_plan): (path, 64-draw chunk) tasks of this batch
	unsigned long long inst_distinct;    // grouped mode: paths that at least one draw uses
	uint32_t inst_detect_bad;            // vgx_tessellate_count: some draw differs from its image in the first period
	uint32_t inst_mismatch;              // vgx_tessellate: the draws no longer repeat with the context's period -> k_flatten_build does the batch
	uint32_t inst_tol_lo_inv;            // ~(smallest bit pattern of tess_tol / scale^2 over the draws), k_inst_tol_range (totals are zeroed: a minimum kept as a maximum)
	uint32_t inst_tol_hi;                // largest one. lo != hi: instances differ in scale -> grouped mode sorts by (path, tolerance class)
	uint32_t inst_tol_varies;            // vgx_tessellate_count, periodic batch: some draw's tolerance differs from its image in the first period
	uint32_t has_general_stroke;         // scan over the meshes: some stroke mesh is not a closed Miter AA / Thin stroke -> k_stroke emits the strokes, else k_stroke_simple
	uint32_t cache_has_uniform;          // vgx_cache_submit: some submitted mesh was cached without per-vertex colours (k_cache_meshes -> k_cache_uniform_colors)
	uint32_t tmpl_bad;                   // vgx_tessellate_count (k_tmpl_check): some draw differs from its image in the first period in a field the
	                                     // flattener or the mesh sizes depend on -> no template mode (vgx_tmpl.hip)
	uint32_t flat_redo;                  // vgx_flatten (vgx_flat1.hip): the first run found degenerate draws -> serial count of the listed draws, second run
	uint32_t has_short_stroke;           // scan over the meshes: some stroke mesh has fewer than VGX_LONG_STROKE elements -> k_stroke emits the general strokes (else k_stroke_long: LDS-staged stores)
	unsigned long long flat_ticket;      // vgx_flatten: next segment (ticket order = output order)
	unsigned long long flat_serial_draws;// vgx_flatten: draws that went through the exact serial builder
	unsigned long long flat_tag;         // vgx_flatten: the batch these totals belong to (VgxF1Args::tag)
	// diagnostics of the first failure (vgx_get_failure_info)
	uint32_t fail_reason;  // VGX_FAIL_*
	uint32_t fail_aux;
	unsigned long long fail_segment;
	// -DVGX_INST_PROFILE builds only: wave clock ticks (100 MHz) summed over all waves per phase of k_flatten_inst
	unsigned long long prof[16];
};
enum {
	VGX_FAIL_NONE = 0,
	VGX_FAIL_SEG_DRAWS = 1,      // more draws in a segment than the table holds (aux = draws)
	VGX_FAIL_SEG_SUBRECS = 2,    // more mesh-producing sub-paths (aux = records)
	VGX_FAIL_SEG_MESHES = 3,     // more meshes (aux = meshes)
	VGX_FAIL_HEAP = 4,           // polyline heap exhausted (aux = vertices wanted)
	VGX_FAIL_OUT_CAPACITY = 5,   // caller's vertex / index / mesh capacity exceeded (aux: 1 vertices, 2 indices, 4 meshes)
	VGX_FAIL_SEG_TABLE = 6,      // more segments than the segment tables hold
	VGX_FAIL_SERIAL_HEAP = 7,
	VGX_FAIL_SERIAL_SUBRECS = 8,
	VGX_FAIL_LOOKBACK_TIMEOUT = 9,
	VGX_FAIL_AGG_RANGE = 10,
	VGX_FAIL_MESH_TOO_LARGE = 11
};
#define VGX_LONG_SUBPATH 2048

// ---- template mode (vgx_tmpl.hip): one period of an instanced batch flattened once, in local space -----------------------
struct VgxTmplMesh // one mesh of the template. 64 bytes
{
	uint32_t poly_first; // first vertex of its polyline in the template's LOCAL polyline
	uint32_t n;          // polyline vertices = elements
	uint32_t v_off;      // first output vertex inside one instance
	uint32_t i_off;      // first output index inside one instance
	uint32_t drawk;      // draw inside the period
	uint32_t kind;       // VgxMeshDesc::kind word
	float f0, f1;        // fills: fringe / 2 (the sign is per instance), -; strokes: hsw, hswAA (thin: fringe, fringe)
	float l0[2], l1[2], l2[2]; // its first three LOCAL vertices: the fill orientation (stroker.cpp:721-723) is the sign of their transformed triangle
	                           // (Round-join stroke meshes: l2[0] = the arc step da, l2[1] = bits of the mesh's first element among the Round-join elements)
	uint32_t pad[2];     // [0]: the draw's fringe (bits); [1]: Round-join stroke meshes: the mesh's number among the instance's Round-join meshes + 1, else 0
};
struct VgxTmplElem // one element (polyline vertex j of template mesh `mesh`), in processing order. 16 bytes
{
	uint32_t mesh;
	uint32_t jq;   // j | (position of the element inside its tile, in OUTPUT order) << 16
	float lx, ly;  // its LOCAL vertex (= local polyline[mesh.poly_first + j]): one record is all an element reads from memory
};
struct VgxTmplRoundMesh // Round-join stroke meshes of the template, in mesh order. [count + 1]; the last entry: mesh = ~0, elem0 = all their elements
{
	uint32_t mesh;   // template mesh
	uint32_t elem0;  // its first element among the instance's Round-join elements
};
struct VgxTmplMeshPlace // Round-join templates, per step: where one mesh of one instance lies in the batch's output. 32 bytes
{
	unsigned long long v, i; // first vertex / index
	uint32_t nv, ni;         // vertices / indices
	uint32_t pad[2];
};
struct VgxTmplTile // one tile of an instance's element stream = one workgroup of k_tmpl_emit. 32 bytes
{
	uint32_t mesh0;     // template mesh that owns the tile's first element; bit 31: that element is the mesh's element 0
	uint32_t mesh_last; // last mesh with an element in the tile
	uint32_t draw0;     // draws of the period whose records the tile reads (and verifies): [draw0, draw0 + ndraws)
	uint32_t ndraws;
	uint32_t nel;       // elements in the tile (the last tile of a class is short)
	uint32_t cmesh0;    // first template mesh of the tile's class (mesh numbers inside an instance = mesh - cmesh0)
	uint32_t cdraw0;    // first saved draw record of the tile's class (= class * period)
	uint32_t pad;
};
// A batch may repeat its period in a FEW flavours ("classes": the same drawing at a handful of scales, say): every instance equals
// one of the class representatives in every field the template depends on. The classes' templates are built as ONE template over
// the concatenated representatives (period * classes draws); this table says where each class lies in it. [classes + 1] entries,
// the last one = the totals.
struct VgxTmplClass // 40 bytes
{
	uint64_t elem0;  // first element (output order of the concatenated representatives)
	uint64_t v0, i0; // first output vertex / index
	uint32_t mesh0;  // first mesh
	uint32_t tile0;  // first tile
	uint32_t pad[2];
};
struct VgxTmplInst // multi-class batches: where instance k's output lies, and its class. 40 bytes
{
	uint64_t v, i;   // first output vertex / index of the instance (templates with Round joins: the per-step table iplace instead)
	uint32_t m;      // first mesh of the instance in the batch's mesh sequence
	uint32_t cls;
	uint32_t cmesh0; // first template mesh of its class
	uint32_t pad;
	uint64_t rel;    // Round-join templates: relem + rel + (an element's number among the TEMPLATE's Round-join elements) = the element's table word
	                 // (= the instance's first word minus its class's first element number; wraps, the sum does not)
};
#define VGX_TMPL_MAX_CLASSES 64

// Capacities the device-side checks compare against.
struct VgxCaps
{
	uint64_t cmd_instances;
	uint64_t poly_vertices;
	uint64_t subpaths;
	uint64_t meshes;
	uint64_t vertices;
	uint64_t indices;
};

#endif
