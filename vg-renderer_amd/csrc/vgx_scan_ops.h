// vgx_scan_ops.h -- the scan operators of the tessellation pipeline (vgx_scan.h OP interface): command instances per draw,
// polyline / sub-path / mesh counts per draw, element / vertex / index offsets per mesh. Shared by the host-side
// launchers (vgx_api.hip) and the single-workgroup kernels of frame-sized batches (vgx_flatten.hip). Device only.
#ifndef VGX_SCAN_OPS_H
#define VGX_SCAN_OPS_H

#include <stddef.h>
#include "vgx_internal.h"
#include "vgx_scan.h"

namespace {

__device__ __forceinline__ void set_status(VgxTotals* t, uint32_t err)
{
	atomicCAS(&t->status, (uint32_t)VGX_OK, err);
}

// ---- scan operators ---------------------------------------------------------------------------------
struct OpCmdPrefix // command instances per draw -> cmd_prefix
{
	const vgx_draw* draws;
	const uint32_t* pathCmdBegin;
	uint32_t npaths;
	uint64_t ndraws;
	uint64_t* prefix;
	VgxTotals* totals;
	uint64_t cap;
	const uint32_t* pathSubBegin; // [npaths + 1] sub-path ending commands per path (static)
	uint64_t* subPrefix;          // [ndraws + 1] out: exclusive scan of the draws' static sub-path counts (dense sub-path records of k_flatten_inst)
	uint32_t period; // instanced batch (vgx_inst.hip): the context expects draws[i].path == draws[i % period].path; 0 = no check
	__device__ uint64_t size() const { return ndraws; }
	__device__ Sum3 load(uint64_t i) const
	{
		Sum3 r = sum3_zero();
		// The whole 64-byte record in 16-byte loads issued together, and the checks below as ONE expression without
		// short-circuit evaluation: written field by field with `&&`, every field became its own load -> s_waitcnt vmcnt(0) ->
		// branch, a chain of twelve dependent memory round trips per draw (the scan over 2.2 M draws: 0.165 -> 0.12 ms).
		static_assert(sizeof(vgx_draw) == 64 && offsetof(vgx_draw, path) == 0 && offsetof(vgx_draw, stroke_flags) == 12 && offsetof(vgx_draw, stroke_width) == 20
			&& offsetof(vgx_draw, scale) == 24 && offsetof(vgx_draw, tess_tol) == 28 && offsetof(vgx_draw, fringe) == 32 && offsetof(vgx_draw, mtx) == 36,
			"the 16-byte loads below pick vgx_draw's fields by position");
		const uint4* q = (const uint4*)(draws + i); // `draws` is 16-byte aligned (include/vgx.h)
		const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
		const uint32_t pimg = period ? draws[i % period].path : 0u;
		const uint32_t p = q0.x;          // path
		const uint32_t sf = q0.w;         // stroke_flags
		if (period && pimg != p) { totals->inst_mismatch = 1u; } // k_flatten_build does this batch
		if (p >= npaths || ((sf & VGX_STROKE_ENABLE) && (VGX_STROKE_CAP(sf) > 2u || VGX_STROKE_JOIN(sf) > 2u))) {
			set_status(totals, VGX_E_INVALID_ARG);
			return r;
		}
		// The draw records are device memory the host never sees: reject parameters that would make the subdivision
		// (flat iff d23^2 <= tol / scale^2 * len^2, path.cpp:105-116) run to the limits of float -- NaN / Inf / zero /
		// negative scale or tolerance, a tolerance below 1e-12 of a unit -- instead of spending hours on them (the
		// reference loops forever on NaN, path.cpp:109). Comparisons are written so that NaN fails them.
		{
			const float sw = __uint_as_float(q1.y), sc = __uint_as_float(q1.z), tt = __uint_as_float(q1.w), fr = __uint_as_float(q2.x);
			const float m0 = __uint_as_float(q2.y), m1 = __uint_as_float(q2.z), m2 = __uint_as_float(q2.w);
			const float m3 = __uint_as_float(q3.x), m4 = __uint_as_float(q3.y), m5 = __uint_as_float(q3.z);
			const float big = 3.0e38f;
			bool ok = (sc > 0.0f) & (sc < big) & (tt > 0.0f) & (tt < big) & (fr >= 0.0f) & (fr < big) & (sw >= 0.0f) & (sw < big);
			ok = ok & (tt / (sc * sc) >= 1.0e-12f);
			ok = ok & (m0 > -big) & (m0 < big) & (m1 > -big) & (m1 < big) & (m2 > -big) & (m2 < big);
			ok = ok & (m3 > -big) & (m3 < big) & (m4 > -big) & (m4 < big) & (m5 > -big) & (m5 < big);
			if (!ok) {
				set_status(totals, VGX_E_NONFINITE);
				return r;
			}
		}
		r.a = pathCmdBegin[p + 1] - pathCmdBegin[p];
		r.b = pathSubBegin[p + 1] - pathSubBegin[p];
		return r;
	}
	__device__ void store(uint64_t i, Sum3 e) const { prefix[i] = e.a; subPrefix[i] = e.b; }
	// A periodic batch that passed the check needs the prefixes of its first period only (+ entry `period` = the sums over one
	// period): k_flatten_inst and k_flatten_gather derive every other draw's in closed form (vgx_sub_prefix_at).
	__device__ uint64_t apply_size() const { return (period && totals->inst_mismatch == 0u) ? (uint64_t)period + 1 : ndraws; }
	__device__ void finish(Sum3 t) const
	{
		prefix[ndraws] = t.a;
		subPrefix[ndraws] = t.b;
		totals->sizes.num_cmd_instances = t.a;
		if (t.a > cap) { set_status(totals, VGX_E_NOSPACE); }
	}
};

struct OpDrawInfo // per-draw polyline / sub-path / mesh counts -> first_* fields
{
	vgx_draw_info* dinfo;
	uint64_t ndraws;
	VgxTotals* totals;
	VgxCaps caps;
	int keepPolyBase; // BUILD mode: first_poly_vertex already holds the draw's heap position
	__device__ uint64_t size() const { return totals->status == VGX_OK ? ndraws : 0; }
	__device__ Sum3 load(uint64_t i) const
	{
		Sum3 r;
		const vgx_draw_info* d = dinfo + i;
		r.a = d->num_poly_vertices; r.b = d->num_subpaths; r.c = d->num_meshes; r.d = d->flags & 1u;
		return r;
	}
	__device__ void store(uint64_t i, Sum3 e) const
	{
		vgx_draw_info* d = dinfo + i;
		if (!keepPolyBase) { d->first_poly_vertex = e.a; }
		d->first_subpath = e.b; d->first_mesh = e.c;
	}
	__device__ void finish(Sum3 t) const
	{
		totals->sizes.num_poly_vertices = t.a;
		totals->sizes.num_subpaths = t.b;
		totals->sizes.num_meshes = t.c;
		totals->sizes.num_serial_draws = t.d;
		if ((!keepPolyBase && t.a > caps.poly_vertices) || t.b > caps.subpaths || t.c > caps.meshes) { set_status(totals, VGX_E_NOSPACE); }
	}
};

// vgx_partition: predicted output vertices per draw = polyline vertices (count pass) x output vertices per polyline vertex of
// the draw's fill / stroke flavour (convexFill 1, convexFillAA 2, stroke 2, strokeAA 4, strokeAAThin 3; Round joins / caps add
// a little that the prediction ignores).
struct OpPartWeight
{
	const vgx_draw* draws;
	const vgx_draw_info* dinfo;
	uint64_t ndraws;
	uint64_t* prefix; // [ndraws + 1]
	__device__ uint64_t size() const { return ndraws; }
	__device__ Sum3 load(uint64_t i) const
	{
		Sum3 r = sum3_zero();
		const uint32_t ff = draws[i].fill_flags, sf = draws[i].stroke_flags;
		uint32_t f = 0;
		if (ff & VGX_FILL_ENABLE) { f += (ff & VGX_FILL_AA) ? 2u : 1u; }
		if (sf & VGX_STROKE_ENABLE) { f += !(sf & VGX_STROKE_AA) ? 2u : ((sf & VGX_STROKE_THIN) ? 3u : 4u); }
		r.a = (uint64_t)dinfo[i].num_poly_vertices * f + 1; // + 1: empty draws still cost a record each, and every range gets a positive weight
		return r;
	}
	__device__ void store(uint64_t i, Sum3 e) const { prefix[i] = e.a; }
	__device__ void finish(Sum3 t) const { prefix[ndraws] = t.a; }
};

// One scan over the meshes for everything the emit kernels need: element offsets (convex fills / polyline strokes
// have separate streams) and vertex / index offsets. Field d carries the index sum in its low 48 bits and the number
// of meshes with more than 65536 vertices above them (a batch cannot hold 2^48 indices: positions are 32-bit counted).
#define VGX_IDX_SUM_MASK ((1ull << 48) - 1)
struct OpMeshAll
{
	const VgxMeshDesc* mdesc;
	vgx_mesh* mtab;
	vgx_mesh* meshesOut; // caller's mesh table, written here when the emit follows at once (vgx_tessellate); else null
	uint64_t* prefixFill;
	uint64_t* prefixStroke;
	VgxTotals* totals;
	VgxCaps caps;
	int checkCaps;
	int fixedSize;       // 1: the item count is fixedCount (single-workgroup callers read it once, uniformly, with a fresh load:
	uint64_t fixedCount; // every thread must see the same count or the block barriers inside the scan mismatch)
	__device__ uint64_t size() const { return fixedSize ? fixedCount : (totals->status == VGX_OK ? totals->sizes.num_meshes : 0); }
	__device__ Sum3 load(uint64_t i) const
	{
		Sum3 r = sum3_zero();
		// all loads first: the conditional store below would otherwise sit between them (possible alias) and turn two
		// parallel loads into a dependent chain (measured: +0.035 ms over 4.35 M meshes)
		const uint32_t kindWord = mdesc[i].kind, polyN = mdesc[i].poly_n;
		const uint32_t nv = mtab[i].num_vertices, nidx = mtab[i].num_indices;
		if (VGX_MD_KIND(kindWord) >= VGX_MESH_STROKE) {
			r.b = polyN;
			// which stroke kernel emits this batch (vgx_stroke.hip): a plain store, at most once per mesh that is not "simple"
			// (closed, Miter join, AA or Thin: bits 0-7 kind, 8 closed, 11-12 join)
			const uint32_t key = kindWord & 0x19FFu;
			const bool simple = key == (0x100u | VGX_MESH_STROKE_AA) || key == (0x100u | VGX_MESH_STROKE_AA_THIN);
			if (!simple) { totals->has_general_stroke = 1u; }
			if (polyN < VGX_LONG_STROKE) { totals->has_short_stroke = 1u; }
		} else { r.a = polyN; }
		r.c = nv;
		r.d = (uint64_t)nidx + (nv > 65536u ? (1ull << 48) : 0ull);
		return r;
	}
	__device__ void store(uint64_t i, Sum3 e) const
	{
		// the record is read BEFORE the stores below (a load behind stores waits for them: vmcnt counts both, in order)
		const bool toCaller = meshesOut && i < caps.meshes;
		vgx_mesh r;
		if (toCaller) { r = mtab[i]; }
		prefixFill[i] = e.a; prefixStroke[i] = e.b;
		mtab[i].first_vertex = e.c; mtab[i].first_index = e.d & VGX_IDX_SUM_MASK;
		if (toCaller) { // the caller's table = the internal one, in the same pass (no k_copy_meshes)
			r.first_vertex = e.c; r.first_index = e.d & VGX_IDX_SUM_MASK;
			meshesOut[i] = r;
		}
	}
	__device__ void finish(Sum3 t) const
	{
		const uint64_t n = size();
		prefixFill[n] = t.a;
		prefixStroke[n] = t.b;
		totals->sizes.num_elements = t.a + t.b;
		totals->sizes.num_fill_elements = t.a;
		totals->sizes.num_vertices = t.c;
		totals->sizes.num_indices = t.d & VGX_IDX_SUM_MASK;
		if (t.d >> 48) { set_status(totals, VGX_E_MESH_TOO_LARGE); }
		if (checkCaps && (t.c > caps.vertices || (t.d & VGX_IDX_SUM_MASK) > caps.indices || totals->sizes.num_meshes > caps.meshes)) { set_status(totals, VGX_E_NOSPACE); }
	}
};

} // namespace

#endif
