// vgx_internal.h -- device-side data layout shared by the kernels and the C-ABI implementation.
#ifndef VGX_INTERNAL_H
#define VGX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vgx.h"
#include "vgx_lane.h"

#include "vgx_internal_types.h"

struct VgxFlattenArgs
{
	VgxPathSetDev ps;
	const vgx_draw* draws;
	uint64_t ndraws;
	const uint64_t* cmd_prefix; // [ndraws+1]
	const uint64_t* sub_prefix; // [ndraws+1] exclusive scan of the draws' static sub-path counts: the flatten kernels of vgx_tessellate store their sub-path records
	                            // densely, record j of draw d at sub_rec[sub_prefix[d] + j]
	uint32_t* cmd_cnt;          // [num_cmd_instances]
	vgx_draw_info* dinfo;       // [ndraws]
	float* poly;                // emit: [cap][2]
	vgx_subpath* subs;          // emit
	VgxMeshDesc* mdesc;         // emit (may be null: flatten-only API)
	VgxMeshPrep* mprep;         // single-pass pipeline: per-mesh constants written together with mdesc (else null: k_mesh_prepare does it)
	vgx_mesh* mtab;             // emit: closed-form mesh sizes are written together with mdesc
	VgxTotals* totals;
	VgxCaps caps;
	int apply_transform;
	// BUILD mode (single-pass flatten of vgx_tessellate): per-sub-path records stored sparsely at the command-instance
	// index of the sub-path's last command
	VgxSubRec* sub_rec;            // [>= static sub-paths of the batch] record j of draw d at sub_prefix[d] + j (BUILD mode)
	int build_mode;                // k_flatten_serial<count>: allocate the draw's vertices from the polyline heap
	uint32_t* serial_list;         // BUILD mode: draws for k_flatten_serial (static serial paths + degenerate draws), unordered
	float* leaf_overflow;          // [VGX_BUILD_WAVES][VGX_BUILD_OVERFLOW][64][2] leaves that did not fit the LDS slots
	int pool_walk;                 // k_flatten_build: pooled cubic walk (vgx_walk.h) instead of one cubic per lane
	int thin_static;               // every path of the set is MOVE_TO / LINE_TO / CLOSE only: k_flatten_thin (vgx_thin.h) in k_flatten_build's place
	// instanced batches (vgx_inst.hip): draws[i].path == draws[i % inst_period].path; 0 = not instanced. When set and the
	// device-side check of this call agrees, k_flatten_inst builds the batch and k_flatten_build exits at once.
	uint32_t inst_period;
	uint32_t inst_block;           // vertices per lane-private heap block
	int inst_waves;                // grid of k_flatten_inst
	// grouped mode of k_flatten_inst (draws that share paths in ANY order; null = periodic mode / off): the draws sorted by
	// path (k_inst_hist / k_inst_plan / k_inst_scatter, run by every vgx_tessellate in this mode)
	const uint32_t* inst_order;      // [ndraws] draw indices, grouped by path
	const uint64_t* inst_start;      // [npaths + 1] first entry of every path in inst_order
	const uint64_t* inst_task_start; // [npaths + 1] first task of every path
	const uint32_t* inst_task_path;  // [tasks] path of every task
	// periodic mode, instances of different scales: lane slot k of the instance walk takes instance inst_perm[k] -- the instances
	// sorted by the tolerance class of their first draw, so that the lanes of a wave flatten with (nearly) the same tolerance and
	// stay in lock-step. Any permutation is valid (null = identity): it only decides which instances share a wave.
	const uint32_t* inst_perm;       // [ndraws / inst_period]
};

struct VgxStrokeArgs
{
	const vgx_draw* draws;
	const float* poly;
	const VgxMeshDesc* mdesc;
	const uint64_t* elem_prefix; // [num_meshes+1] exclusive prefix of poly_n over the meshes of this kernel's kind class
	const uint64_t* elem_prefix_fill;   // fills only (strokes contribute 0)
	const uint64_t* elem_prefix_stroke; // strokes only
	VgxMeshPrep* mprep;          // [num_meshes]
	vgx_mesh* mtab;              // [num_meshes] count: writes num_*, scan fills first_*, emit reads
	float* pos;
	uint32_t* color;
	uint16_t* idx;
	vgx_mesh* meshes_out;        // caller's mesh table (emit copies mtab into it)
	const uint32_t* mesh_base;   // assembly armed: vertices in front of each mesh inside its vertex buffer (added to every index); else null
	VgxTotals* totals;
	VgxCaps caps;
	int no_long;                 // frame-sized call: k_stroke takes the strokes whatever their length (k_stroke_long is not launched)
	int tile_mode;               // the host also launches k_emit_tiles (vgx_tile.hip): k_fill leaves the batches that kernel takes alone
};


// template mode (vgx_tmpl.hip)
#ifndef VGX_TMPL_THREADS
#define VGX_TMPL_THREADS 512   /* threads per workgroup of k_tmpl_emit */
#endif
#ifndef VGX_TMPL_MAX_TILE
#define VGX_TMPL_MAX_TILE 2048 /* elements per tile = per workgroup (what its LDS stages hold); a multiple of VGX_TMPL_THREADS. Tiger x10k, same
                                * box: 256 threads x 1024 elements 2.18 ms, 512 x 2048 2.06, 512 x 3072 2.15, 1024 x 4096 2.15, 128 x 1024 2.26 */
#endif
struct VgxTmplBuild // count pass: the first period's ordinary count + emit results -> template tables
{
	const vgx_draw* draws;       // the batch's first period
	const VgxMeshDesc* mdesc;
	const VgxMeshPrep* mprep;
	const vgx_mesh* mtab;
	const float2* poly;            // the period's LOCAL polyline (two-phase flatten with apply_transform = 0)
	const uint64_t* prefix_fill;   // [num_meshes + 1]
	const uint64_t* prefix_stroke; // [num_meshes + 1]
	uint64_t num_meshes, num_elems;
	uint32_t tile;               // elements per tile of the processing order (a multiple of 64)
	VgxTmplMesh* tmesh;
	vgx_mesh* tmtab;
	VgxTmplElem* telem;
	VgxTmplTile* ttile;          // [tiles]
	uint32_t period;
	uint32_t nclasses;           // the draws are `nclasses` representatives of `period` draws each
	uint64_t num_vertices, num_indices; // output totals of the concatenated representatives
	VgxTmplClass* cls;           // [nclasses + 1], written by the first build kernel
	VgxTmplRoundMesh* trmesh;    // [num_meshes + 1] (used: Round-join meshes + 1)
	uint2* tmsz;                 // [num_meshes] vertices, indices of the mesh (Round joins: 0x80000000 | number among the Round-join meshes, 0)
	uint32_t has_round;          // the template holds Round-join meshes (k_tmpl_styles): number them
	struct Sum3* partial;        // scratch of the device scan
};
struct VgxTmplArgs // one step
{
	const vgx_draw* draws;
	uint64_t ndraws;
	uint64_t ninst;
	uint32_t period;
	uint32_t npaths;
	const vgx_draw* tdraws;      // the first period as vgx_tessellate_count saw it
	const float2* tpoly;         // local polyline of one period
	const VgxTmplMesh* tmesh;
	const vgx_mesh* tmtab;       // mesh records of one instance (offsets relative to the instance)
	const VgxTmplElem* telem;
	const VgxTmplTile* ttile;
	uint32_t tile;               // elements per tile (= one workgroup of k_tmpl_emit)
	uint32_t tiles_per_inst;
	vgx_sizes inst;              // sizes of ONE instance
	float* pos;
	uint32_t* color;
	uint16_t* idx;
	vgx_mesh* meshes_out;        // may be null
	const uint32_t* mesh_base;   // assembly armed: [ninst * meshes] vertices in front of each mesh inside its draw command; else null
	VgxCaps caps;                // vertices / indices / meshes of the caller's buffers
	VgxTotals* totals;
	// several classes (else null / 0): instance k = class iinfo[k].cls, workgroup b = tile wg[b].y (of the concatenated template) of instance wg[b].x
	const VgxTmplInst* iinfo;    // [ninst + 1]; the last entry = the batch totals
	const uint2* wg;             // [num_wg]
	uint64_t num_wg;
	vgx_sizes total;             // sizes of the whole batch
	uint32_t general;            // stroke styles of the template: 0 = closed Miter AA / Thin (k_tmpl_emit), 1 = + open Miter, Butt / Square caps (k_tmpl_emit_open), 2 = any (k_tmpl_emit_general), 3 = + Round joins (k_tmpl_emit_round), 4 = closed Miter + closed Bevel only (k_tmpl_emit_bevel), 5 / 6 = 4 + closed / + open AA strokes with Round joins (k_tmpl_emit_round_aa / _open)
	// Round joins (templates of ONE class): the arc of every join is counted on the instance's TRANSFORMED polyline (stroker.cpp:1146, 1592), so the
	// sizes of those meshes -- and with them every output place behind them -- belong to the instance. Per-step tables, written by
	// vgx_launch_tmpl_round_sizes and read by k_tmpl_emit_round:
	uint32_t num_round;          // Round-join stroke meshes per instance (VgxTmplMesh::pad[1] = the mesh's number among them + 1)
	uint32_t num_round_elems;    // their elements per instance
	const VgxTmplRoundMesh* trmesh; // [num_round + 1]
	const uint2* tmsz;           // [meshes of the template] vertices, indices (Round joins: 0x80000000 | number among the Round-join meshes, 0): what the scan over the meshes reads
	unsigned long long* rsz;     // [ninst * num_round * 2] vertices, indices of every such mesh
	uint2* relem;                // [ninst * num_round_elems] per element of such a mesh: first vertex / index inside the mesh, size and inner side of the element in front (tmpl_round_word)
	VgxTmplMeshPlace* mplace;    // [ninst * meshes] per mesh of the batch: first vertex, first index in the BATCH (iplace set: inside its INSTANCE); vertices, indices
	unsigned long long* itot;    // [ninst * 2] (per-instance shape of the sizes pass only, else null) vertices, indices of the instance
	unsigned long long* iplace;  // [ninst * 2] ... first vertex, first index of the instance in the batch
	// Round joins, several classes (round 6; the per-instance shape of the sizes pass only): class c's Round-join meshes are trmesh[cls[c].pad[1] ..
	// cls[c + 1].pad[1]) (entry [nclasses]: all of them), an instance's tables lie at iinfo[k].m (mplace) and iinfo[k].rel (relem)
	const VgxTmplClass* cls;     // [nclasses + 1] (null: one class)
	uint32_t round_lds;          // Round-join meshes of the largest class (k_tmpl_round_sizes_inst's LDS table)
};
struct Sum3;
bool vgx_tmpl_round_per_instance(const VgxTmplArgs& a); // the sizes pass places the meshes per instance (many instances of a few hundred meshes): the caller sets a.itot / a.iplace
void vgx_launch_tmpl_round_sizes(const VgxTmplArgs& a, Sum3* partial, hipStream_t s); // Round-join templates, in front of vgx_launch_tmpl_emit: the tables above, totals->sizes, VGX_E_NOSPACE against a.caps
void vgx_launch_tmpl_mtab(const VgxTmplArgs& a, vgx_mesh* mtab, VgxMeshDesc* mdesc, hipStream_t s); // assembly armed: the whole batch's mesh table + mesh -> draw
void vgx_launch_tmpl_check(const vgx_draw* draws, uint64_t ndraws, uint32_t npaths, VgxTotals* totals, hipStream_t s); // after vgx_launch_inst_detect
void vgx_launch_tmpl_styles(const VgxTmplBuild& b, hipStream_t s);  // stroke styles of the template -> b.cls[nclasses].pad[0] (zeroed by the caller), needs mdesc only
void vgx_launch_tmpl_classes(const VgxTmplBuild& b, hipStream_t s); // fills b.cls from the representatives' count + emit results
void vgx_launch_tmpl_class_sums(const VgxTmplBuild& b, const vgx_draw_info* dinfo, const uint64_t* cmdPrefix, uint64_t numDraws, const vgx_sizes& all, unsigned long long* sums, hipStream_t s); // after vgx_launch_tmpl_classes: [nclasses + 1][5] prefixes in front of every class
void vgx_launch_tmpl_build(const VgxTmplBuild& b, hipStream_t s);   // after vgx_launch_tmpl_classes
void vgx_launch_tmpl_hash(const vgx_draw* draws, uint64_t ndraws, uint64_t period, unsigned long long* hashes, hipStream_t s); // hashes[instance], zeroed by the caller
void vgx_launch_tmpl_check_cls(const vgx_draw* draws, uint64_t ndraws, uint64_t period, const uint32_t* inst_cls, const uint32_t* cls_rep, VgxTotals* totals, hipStream_t s);
void vgx_launch_tmpl_emit(const VgxTmplArgs& a, hipStream_t s);
#ifndef VGX_TMPL_RC_THREADS
#define VGX_TMPL_RC_THREADS 512 /* k_tmpl_emit_round_aa (templates of closed strokes with Round joins): threads per workgroup, */
#define VGX_TMPL_RC_TILE 2048   /* elements per tile */
#endif
#ifndef VGX_TMPL_RC_WAVES
#define VGX_TMPL_RC_WAVES 6     /* waves per SIMD its registers are limited for (80 VGPRs: three workgroups per CU) */
#endif
#ifndef VGX_TMPL_GENERAL_TILE
#define VGX_TMPL_GENERAL_TILE 2048 /* tile size of templates that hold general strokes (the LDS stages of k_tmpl_emit_general) */
#endif

// merging two mesh sequences of a frame (vgx_merge.hip)
struct VgxMergeArgs
{
	vgx_cache_desc a, b;       // sequence A (vgx_tessellate's meshes) and B (external meshes); both sorted by draw
	const uint32_t* b_draw;    // frame draw of every B mesh (null: the records' own draw fields)
	uint32_t* order;           // [na + nb] merged position -> source mesh (bit 31: from B)
	vgx_mesh* mtab;            // [na + nb] merged mesh table (context scratch; assembly reads it)
	VgxMeshDesc* mdesc;        // [na + nb] only .draw is written (assembly's mesh -> draw)
	vgx_mesh* meshes_out;      // caller's table (may be null)
	float* pos;
	uint32_t* color;
	uint16_t* idx;
	const uint32_t* mesh_base; // assembly armed: index base per merged mesh; else null
	const void* b_uv;          // per-vertex UVs of sequence B (may be null), uv_bytes each
	void* uv_out;              // the armed assembly's UV stream (null: none)
	uint32_t uv_bytes;
	VgxTotals* totals;
	VgxCaps caps;
};
void vgx_launch_merge_rank(const VgxMergeArgs& a, hipStream_t s);
void vgx_launch_merge_scan(const VgxMergeArgs& a, void* partial, hipStream_t s);
void vgx_launch_merge_copy(const VgxMergeArgs& a, hipStream_t s);

// concave-fill fringes (vgx_concave.hip)
struct VgxConcaveArgs
{
	const float* contour_verts;
	const vgx_contour* contours;
	uint64_t ncontours;
	uint64_t num_contour_vertices;
	const vgx_concave_fill* fills;
	uint64_t nfills;
	float* moved;              // vgx_concave_move only
	const float* tess_pos;     // vgx_concave_emit only (and everything below)
	const uint16_t* tess_idx;
	const vgx_mesh* mtab;      // [nfills] after the scan over the fills
	float* pos;
	uint32_t* color;
	uint16_t* idx;
	VgxTotals* totals;
};
void vgx_launch_concave_move(const VgxConcaveArgs& a, hipStream_t s);
void vgx_launch_concave_emit(const VgxConcaveArgs& a, hipStream_t s);

// launchers (defined in the .hip files)
void vgx_launch_flatten(bool emit, const VgxFlattenArgs& a, int numBlocks, hipStream_t s);
void vgx_launch_flatten_build(const VgxFlattenArgs& a, int waves, hipStream_t s, bool serialCount = true);   // single-pass: subdivide once, polyline -> heap
void vgx_launch_flatten_inst(const VgxFlattenArgs& a, int waves, hipStream_t s);  // instanced batches: one lane per instance (vgx_inst.hip)
void vgx_launch_inst_perm(const vgx_draw* draws, uint64_t ninst, uint32_t period, uint32_t nc, uint32_t* classHist /* [nc + 1] */, uint32_t* perm /* [ninst] */, VgxTotals* totals, bool multi /* testing: the several-kernel form for any count */, hipStream_t s);
void vgx_launch_inst_detect(const vgx_draw* draws, uint64_t ndraws, VgxTotals* totals, hipStream_t s); // count pass: period of the path sequence
// grouped mode: histogram of the draws' paths -> per-path ranges and task list (taskPath may be null: counts only) -> draw order
// nc: tolerance classes per path (1 = sort by path only); hist / cursor / keyStart hold npaths * nc + 1 entries (keyStart unused when nc == 1)
void vgx_launch_inst_group(const vgx_draw* draws, uint64_t ndraws, uint32_t npaths, uint32_t nc, uint32_t* hist, uint32_t* cursor, uint64_t* keyStart, uint64_t* start,
	uint64_t* taskStart, uint32_t* taskPath, uint64_t capTasks, uint32_t* order, VgxTotals* totals, void* scanPartial /* Sum3[VGX_SCAN_BLOCKS] */, hipStream_t s);
// frame-sized batches (vgx_flatten.hip): one-workgroup kernels instead of chains of dependent launches; the operators are
// the OpCmdPrefix / OpDrawInfo / OpMeshAll of vgx_scan_ops.h, passed type-erased (the header is device code)
#define VGX_SMALL_DRAWS 2048
struct VgxStrokeArgs;
void vgx_launch_small_front(const void* opCmdPrefix, vgx_draw_info* dinfo, hipStream_t s);
void vgx_launch_small_middle(const VgxFlattenArgs& f, const VgxStrokeArgs& st, const void* opDraws, const void* opMeshes, vgx_sizes* devSizes, uint32_t* devStatus, hipStream_t s);
void vgx_launch_flatten_gather(const VgxFlattenArgs& a, hipStream_t s);  // after the draw scan: ordered mesh descriptors
void vgx_launch_flatten_gather_ordered(const VgxFlattenArgs& a, hipStream_t s); // vgx_tessellate's one-walk route: mesh descriptors from k_flat1's ordered records
#ifndef VGX_BUILD_WAVES
#define VGX_BUILD_WAVES 4096
#endif
#define VGX_BUILD_BLOCK 8192 /* polyline vertices per wave-private heap block */
#define VGX_BUILD_OVERFLOW 120 /* leaves per lane beyond the LDS slots kept in the wave's global overflow area */
void vgx_launch_stroke(bool emit, const VgxStrokeArgs& a, int numBlocks, hipStream_t s);
void vgx_launch_mesh_prepare(const VgxStrokeArgs& a, hipStream_t s);

// draw-command assembly (vgx_assemble.hip)
struct VgxAsmArgs
{
	const vgx_mesh* mtab;      // after the scan over meshes (first_vertex / first_index filled)
	uint32_t* jump0;           // [meshes + 1] next / doubled jump table (ping)
	uint32_t* jump1;           // [meshes + 1] (pong)
	uint32_t* start;           // [cap_start] first mesh of every vertex buffer; cap_start is a power of two
	uint64_t cap_start;
	uint32_t* mesh_base;       // [meshes] out: vertices in front of the mesh inside its vertex buffer
	vgx_drawcmd* drawcmds;     // caller's table
	uint64_t cap_drawcmds;
	uint64_t* dev_num_drawcmds;
	uint32_t max_vb;
	VgxTotals* totals;
	// VGX_ASM_SPLIT_STATE: a change of the draws' state_key between consecutive meshes also starts a draw command
	uint32_t flags;
	const VgxMeshDesc* mdesc;  // mesh -> draw
	const vgx_draw* draws;     // draw -> state_key (null: no draws at this level, e.g. the shape cache)
	uint32_t* mesh_cmd;        // [meshes] scratch: draw command of every mesh (aliases jump0 once the doubling is done)
	void* partial;             // scan partials (Sum3[VGX_SCAN_BLOCKS])
	uint64_t max_meshes;       // capacity of the per-mesh scratch
	uint64_t scan_bound;       // host-known bound of the mesh count (launch shape of the scan over the meshes): the caller's mesh capacity when it gave a table
	// white-pixel UV stream (vg.cpp:5218-5225)
	void* uv; uint32_t uv_bytes; uint32_t uv_value[2];
};
void vgx_launch_assemble(const VgxAsmArgs& a, hipStream_t s);

// shape-cache instancing (vgx_cache.hip)
struct VgxCacheArgs
{
	vgx_cache_desc cache;
	const vgx_cache_instance* inst;
	uint64_t ninst;
	const uint64_t* inst_mesh_prefix; // [ninst + 1] exclusive scans over the instances
	const uint64_t* inst_vert_prefix;
	const uint64_t* inst_idx_prefix;
	vgx_mesh* mtab;                   // output mesh table (context scratch; the assembly step reads it)
	vgx_mesh* meshes_out;             // caller's copy (may be null)
	float* pos;
	uint32_t* color;
	uint16_t* idx;
	const uint32_t* mesh_base;        // assembly armed: index base per output mesh; else null
	VgxTotals* totals;
};
void vgx_launch_cache_localize(const vgx_draw* draws, uint64_t ndraws, float* pos, const vgx_mesh* meshes, uint64_t numMeshes, hipStream_t s);
void vgx_launch_cache_meshes(const VgxCacheArgs& a, hipStream_t s);
void vgx_launch_cache_copy(const VgxCacheArgs& a, int numBlocks, hipStream_t s);
void vgx_launch_fill(const VgxStrokeArgs& a, int numBlocks, hipStream_t s);

#endif
