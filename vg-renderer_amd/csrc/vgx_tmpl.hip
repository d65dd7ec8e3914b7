// vgx_tmpl.hip -- TEMPLATE mode of vgx_tessellate: a drawing submitted for many instances that differ only in their
// transform and colours (the BASELINE headline: Tiger x 10 000).
//
// The reference flattens a path in its LOCAL space (pathXXX, src/path.cpp:62-784, with tessTol = tol / scale^2) and only
// then applies the state transform to the finished polyline (transformPath -> vgutil::batchTransformPositions,
// src/vg.cpp:4957-4975, src/vg_util.cpp:266-272) before the stroker sees it. Instances whose draw records agree in
// everything the flattener and the stroker's SIZES depend on -- path, fill / stroke flags, stroke width, scale,
// tolerance, fringe -- therefore share one local polyline, one set of sub-paths and -- Round joins apart, whose point count
// depends on the transformed geometry (stroker.cpp:1146, 1592): see the end of this comment -- one set of mesh sizes, bit for bit. What differs per
// instance is transformPos2D of every polyline vertex (vg_util.h:24-28) and everything the stroker derives from the
// transformed vertices (directions, extrusion vectors, inner side of every join, fill orientation).
//
// So vgx_tessellate_count flattens the FIRST period of such a batch once -- or, when the instances come in a few flavours
// ("classes": the same drawing at a handful of scales), one representative per class --, with the ordinary two-phase kernels
// and apply_transform = 0, and keeps the result as a template (a few hundred KB per class, L2 resident): local polyline,
// per-mesh records with closed-form output offsets inside one instance, and an element table in processing order. One step of
// vgx_tessellate is then ONE kernel (k_tmpl_emit; k_tmpl_emit_open / k_tmpl_emit_general when the template holds open Miter
// strokes / any other style without Round joins): one workgroup per TILE of one instance's elements -- the instance's draw
// records verified against the saved ones (a mismatch ends the call with VGX_E_STALE; the finiteness checks of the ordinary
// path), template vertices through transformPos2D with the instance's matrix ONCE (staged in LDS), edge directions once
// (staged in LDS), the stroker's per-element arithmetic (strokerConvexFillAA stroker.cpp:713-807; closed Miter polylineStrokeAA
// / AAThin :1524-1579, 1970-1984, 2060-2110, 2295-2306; open and general strokes through tmpl_stroke_elem_open / the element
// code of vgx_elem.h), stores at closed-form addresses -- no polyline heap, no scans, no mesh descriptors in HBM. Inside a tile
// the fill elements come first, then the stroke elements, so a wave's chunks are (almost) pure and the two mesh kinds of a
// draw, which interleave in the output streams, are written by the same workgroup within microseconds (the hole pattern
// DESIGN.md section 9 measured at 2.4 TB/s when two kernels wrote them milliseconds apart).
// Results are identical to the ordinary path's by construction and by test (tests/test_gpu_tmpl.py: VGX_TMPL=0 vs 1
// byte for byte, both against the reference).
// Round 5: (1) Round joins. The sizes of such meshes -- and every output place behind them -- belong to the instance: a step first counts
// them (k_tmpl_round_sizes*: the emit kernel's own functions on the same inputs, a per-element table of places), places meshes and instances
// by scans (capacities checked on the device) and emits with those places: k_tmpl_emit_round_aa[_open] (AA strokes with Round joins beside
// closed Miter / Bevel ones: tmpl_stroke_elem_round) or k_tmpl_emit_round (the general body). (2) Closed Bevel strokes have a routine and,
// when the template holds nothing else, a kernel of their own (tmpl_stroke_elem_bevel, k_tmpl_emit_bevel). (3) Static batches
// (vgx_set_static_batches): a draw list WITHOUT a period is a template of ONE instance -- the same kernels, the tables in HBM instead of L2.
#include <stddef.h>
#include <type_traits>
#include "vgx_internal.h"
#include "vgx_wave.h"
#include "vgx_elem.h"
#include "vgx_tmpl_elem.h"
#include "vgx_scan_ops.h"

namespace {

// ---- count pass: is the batch a template batch? ---------------------------------------------------------------------
// Bit patterns of the fields the flattener and the mesh sizes depend on; colours, transform and state_key may differ.
static_assert(sizeof(vgx_draw) == 64 && offsetof(vgx_draw, path) == 0 && offsetof(vgx_draw, fill_flags) == 4 && offsetof(vgx_draw, fill_color) == 8
	&& offsetof(vgx_draw, stroke_flags) == 12 && offsetof(vgx_draw, stroke_color) == 16 && offsetof(vgx_draw, stroke_width) == 20 && offsetof(vgx_draw, scale) == 24
	&& offsetof(vgx_draw, tess_tol) == 28 && offsetof(vgx_draw, fringe) == 32 && offsetof(vgx_draw, mtx) == 36, "draw records are read as four 16-byte words");
__device__ __forceinline__ bool tmpl_same(const uint4 a0, const uint4 a1, const uint4 a2, const uint4 b0, const uint4 b1, const uint4 b2)
{
	return ((a0.x == b0.x) & (a0.y == b0.y) & (a0.w == b0.w) & (a1.y == b1.y) & (a1.z == b1.z) & (a1.w == b1.w) & (a2.x == b2.x)) != 0;
}

// the checks of OpCmdPrefix::load (vgx_scan_ops.h) on one draw record
__device__ __forceinline__ uint32_t tmpl_validate(const uint4 q0, const uint4 q1, const uint4 q2, const uint4 q3, uint32_t npaths)
{
	const uint32_t sf = q0.w;
	if (q0.x >= npaths || ((sf & VGX_STROKE_ENABLE) && (VGX_STROKE_CAP(sf) > 2u || VGX_STROKE_JOIN(sf) > 2u))) { return VGX_E_INVALID_ARG; }
	const float sw = __uint_as_float(q1.y), sc = __uint_as_float(q1.z), tt = __uint_as_float(q1.w), fr = __uint_as_float(q2.x);
	const float m0 = __uint_as_float(q2.y), m1 = __uint_as_float(q2.z), m2 = __uint_as_float(q2.w);
	const float m3 = __uint_as_float(q3.x), m4 = __uint_as_float(q3.y), m5 = __uint_as_float(q3.z);
	const float big = 3.0e38f;
	bool ok = (sc > 0.0f) & (sc < big) & (tt > 0.0f) & (tt < big) & (fr >= 0.0f) & (fr < big) & (sw >= 0.0f) & (sw < big);
	ok = ok & (tt / (sc * sc) >= 1.0e-12f);
	ok = ok & (m0 > -big) & (m0 < big) & (m1 > -big) & (m1 < big) & (m2 > -big) & (m2 < big);
	ok = ok & (m3 > -big) & (m3 < big) & (m4 > -big) & (m4 < big) & (m5 > -big) & (m5 < big);
	return ok ? (uint32_t)VGX_OK : (uint32_t)VGX_E_NONFINITE;
}

// After k_inst_find (vgx_inst.hip): P = first repetition of draws[0].path. Every draw against its image in the first period.
__global__ __launch_bounds__(256) void k_tmpl_check(const vgx_draw* draws, uint64_t ndraws, uint32_t npaths, VgxTotals* totals)
{
	const unsigned long long inv = totals->inst_detect_inv;
	const unsigned long long P = ~0ull - inv;
	if (inv == 0 || ndraws % P != 0 || P > 0x7FFFFFFFull) {
		if (blockIdx.x == 0 && threadIdx.x == 0) { totals->tmpl_bad = 1u; }
		return;
	}
	bool bad = false;
	uint32_t err = VGX_OK;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t k = i % P;
	const uint64_t kstep = stride % P;
	for (; i < ndraws; i += stride) {
		const uint4* q = (const uint4*)(draws + i);
		const uint4* t = (const uint4*)(draws + k);
		const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
		const uint4 t0 = t[0], t1 = t[1], t2 = t[2];
		bad = bad || !tmpl_same(q0, q1, q2, t0, t1, t2);
		const uint32_t e = tmpl_validate(q0, q1, q2, q3, npaths);
		if (e != VGX_OK && err == VGX_OK) { err = e; }
		k += kstep;
		if (k >= P) { k -= P; }
	}
	if (bad) { totals->tmpl_bad = 1u; }
	if (err != VGX_OK) { set_status(totals, err); }
}

// ---- count pass: several classes? -------------------------------------------------------------------------------------
// hashes[instance] = sum over the instance's draws of a hash of (position in the period, the template fields): instances with the
// same hash are CANDIDATES for one class; k_tmpl_check_cls then compares every draw with its class representative bit by bit.
__device__ __forceinline__ unsigned long long tmpl_mix(unsigned long long h, uint32_t v)
{
	h ^= v; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29;
	return h;
}
__global__ __launch_bounds__(256) void k_tmpl_hash(const vgx_draw* draws, uint64_t ndraws, uint64_t P, unsigned long long* hashes)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndraws; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint4* q = (const uint4*)(draws + i);
		const uint4 q0 = q[0], q1 = q[1], q2 = q[2];
		const uint64_t inst = i / P;
		unsigned long long h = tmpl_mix(0x243F6A8885A308D3ull, (uint32_t)(i - inst * P));
		h = tmpl_mix(h, q0.x); h = tmpl_mix(h, q0.y); h = tmpl_mix(h, q0.w); h = tmpl_mix(h, q1.y); h = tmpl_mix(h, q1.z); h = tmpl_mix(h, q1.w); h = tmpl_mix(h, q2.x);
		atomicAdd(&hashes[inst], h);
	}
}
__global__ __launch_bounds__(256) void k_tmpl_check_cls(const vgx_draw* draws, uint64_t ndraws, uint64_t P, const uint32_t* inst_cls, const uint32_t* cls_rep, VgxTotals* totals)
{
	bool bad = false;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndraws; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t inst = i / P;
		const uint64_t k = i - inst * P;
		const uint4* q = (const uint4*)(draws + i);
		const uint4* t = (const uint4*)(draws + (uint64_t)cls_rep[inst_cls[inst]] * P + k);
		bad = bad || !tmpl_same(q[0], q[1], q[2], t[0], t[1], t[2]);
	}
	if (bad) { totals->tmpl_bad = 1u; }
}

// ---- count pass: the template's tables from the representatives' ordinary count + emit ------------------------------
// (one representative = the batch's first period for a batch of one class)
// Where each class lies in the concatenated template. One thread: the class count is tiny.
__global__ void k_tmpl_classes(VgxTmplBuild B)
{
	const uint64_t M = B.num_meshes;
	uint32_t tile0 = 0;
	for (uint32_t c = 0; c <= B.nclasses; ++c) {
		// first mesh whose draw belongs to class c or a later one (meshes are in draw order)
		uint64_t lo = 0, hi = M;
		const uint64_t d0 = (uint64_t)c * B.period;
		while (lo < hi) {
			const uint64_t mid = (lo + hi) >> 1;
			if (B.mdesc[mid].draw < d0) { lo = mid + 1; } else { hi = mid; }
		}
		VgxTmplClass r;
		r.mesh0 = (uint32_t)lo;
		r.elem0 = B.prefix_fill[lo] + B.prefix_stroke[lo];
		r.v0 = lo < M ? B.mtab[lo].first_vertex : B.num_vertices;
		r.i0 = lo < M ? B.mtab[lo].first_index : B.num_indices;
		r.tile0 = tile0;
		r.pad[0] = B.cls[c].pad[0]; r.pad[1] = 0; // (entry [nclasses]: the style bits k_tmpl_styles left there)
		B.cls[c] = r;
		if (c > 0) {
			const uint64_t e = r.elem0 - B.cls[c - 1].elem0;
			B.cls[c].tile0 = B.cls[c - 1].tile0 + (uint32_t)((e + B.tile - 1) / B.tile);
		}
		tile0 = B.cls[c].tile0;
	}
}
// Per class what the concatenated count holds in front of its first draw (round 6: the classes used to go through the count pipeline one
// by one for their sizes -- eighteen pipelines of three host round trips each for the Tiger at seven scales; the sizes of a class are
// differences of prefixes the ONE concatenated run has already made). sums[c] = { polyline vertices, sub-paths, command instances, fill
// elements, serial draws } in front of class c; entry [nclasses] = the totals. One workgroup per entry.
__global__ __launch_bounds__(256) void k_tmpl_class_sums(VgxTmplBuild B, const vgx_draw_info* dinfo, const uint64_t* cmdPrefix, uint64_t numDraws, vgx_sizes all, unsigned long long* sums)
{
	__shared__ unsigned long long s_serial;
	const uint32_t c = blockIdx.x;
	const uint64_t d0 = (uint64_t)c * B.period;
	if (threadIdx.x == 0) { s_serial = 0ull; }
	__syncthreads();
	// serial draws IN FRONT of the class: counted per class, summed by the host (entry c holds class c - 1's count, entry 0 none)
	if (c > 0) {
		unsigned long long n = 0;
		for (uint64_t d = d0 - B.period + threadIdx.x; d < d0; d += 256) { n += (dinfo[d].flags & 1u) ? 1ull : 0ull; }
		if (n) { atomicAdd(&s_serial, n); }
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long* o = sums + 5ull * c;
		const bool end = d0 >= numDraws;
		o[0] = end ? all.num_poly_vertices : dinfo[d0].first_poly_vertex;
		o[1] = end ? all.num_subpaths : dinfo[d0].first_subpath;
		o[2] = cmdPrefix[end ? numDraws : d0];
		o[3] = B.prefix_fill[B.cls[c].mesh0];
		o[4] = s_serial;
	}
}
__device__ __forceinline__ uint32_t tmpl_class_of_draw(const VgxTmplBuild& B, uint32_t draw) { return draw / B.period; }

// which stroke styles the template holds -> cls[nclasses].pad[0]: bit 0 = open Miter strokes with Butt / Square caps, bit 1 = any other
// stroke that is not closed Miter AA / Thin, bit 2 = Round joins, bit 3 = closed Bevel AA / Thin strokes (the host zeroes the table first;
// k_tmpl_classes keeps the word)
__device__ __forceinline__ bool tmpl_stroke_is_open_fast(uint32_t kindWord);
__device__ __forceinline__ bool tmpl_stroke_is_closed_bevel(uint32_t kindWord);
__device__ __forceinline__ bool tmpl_stroke_is_closed_round_aa(uint32_t kindWord); // bit 4
// the meshes whose sizes depend on the transformed geometry: Round joins count their arc points there (stroker.cpp:1146, 1592);
// thin strokes turn Round joins into Bevel ones (:318-327)
__device__ __forceinline__ bool tmpl_is_round(uint32_t kindWord)
{
	const uint32_t kind = VGX_MD_KIND(kindWord);
	return (kind == VGX_MESH_STROKE || kind == VGX_MESH_STROKE_AA) && VGX_MD_JOIN(kindWord) == VGX_JOIN_ROUND;
}
#ifndef VGX_TMPL_MAXM
#define VGX_TMPL_MAXM 160      /* meshes / draws per tile the LDS tables hold; a tile that needs more takes the per-lane fallback */
#endif
__global__ __launch_bounds__(256) void k_tmpl_styles(VgxTmplBuild B)
{
	uint32_t f = 0;
	for (uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m < B.num_meshes; m += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t kw = B.mdesc[m].kind;
		const uint32_t kind = VGX_MD_KIND(kw);
		if (kind >= VGX_MESH_STROKE && !stroke_elem_is_simple(kind, VGX_MD_CLOSED(kw) != 0, VGX_MD_JOIN(kw))) {
			f |= tmpl_stroke_is_open_fast(kw) ? 1u : (tmpl_stroke_is_closed_bevel(kw) ? 8u : ((VGX_MD_KIND(kw) == VGX_MESH_STROKE_AA && VGX_MD_JOIN(kw) == VGX_JOIN_ROUND) ? (VGX_MD_CLOSED(kw) ? 16u : 32u) : 2u)); // (bits 4 / 5: closed / open AA strokes with Round joins)
		}
		if (tmpl_is_round(kw)) { f |= 4u; }
	}
	if (f) { atomicOr(&B.cls[B.nclasses].pad[0], f); }
}

__global__ __launch_bounds__(256) void k_tmpl_meshes(VgxTmplBuild B)
{
	const uint64_t M = B.num_meshes;
	for (uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (uint64_t)gridDim.x * blockDim.x) {
		const VgxMeshDesc md = B.mdesc[m];
		const VgxMeshPrep pr = B.mprep[m];
		vgx_mesh mt = B.mtab[m];
		const uint32_t ci = tmpl_class_of_draw(B, md.draw);
		const VgxTmplClass cl = B.cls[ci];
		mt.first_vertex -= cl.v0; mt.first_index -= cl.i0; mt.draw -= ci * B.period; // relative to ONE instance of the class
		VgxTmplMesh t;
		t.poly_first = (uint32_t)md.poly_first;
		t.n = md.poly_n;
		t.v_off = (uint32_t)mt.first_vertex;
		t.i_off = (uint32_t)mt.first_index;
		t.drawk = mt.draw;
		t.kind = md.kind;
		if (VGX_MD_KIND(md.kind) >= VGX_MESH_STROKE) { t.f0 = pr.f0; t.f1 = pr.f1; } // hsw / hswAA (thin: fringe, fringe)
		else { t.f0 = B.draws[md.draw].fringe * 0.5f; t.f1 = 0.0f; }                // |aa| = fringe / 2 (stroker.cpp:723); the sign is per instance
		const float2* v = B.poly + md.poly_first;
		const float2 a = v[0], b = v[md.poly_n > 1 ? 1 : 0], c = v[md.poly_n > 2 ? 2 : 0];
		t.l0[0] = a.x; t.l0[1] = a.y; t.l1[0] = b.x; t.l1[1] = b.y; t.l2[0] = c.x; t.l2[1] = c.y;
		t.pad[0] = __float_as_uint(pr.f2); t.pad[1] = 0; // the draw's fringe (general strokes: Butt-cap fringes, thin strokes)
		B.tmesh[m] = t;
		B.tmtab[m] = mt;
		B.tmsz[m] = make_uint2(mt.num_vertices, mt.num_indices);
	}
}

// Round-join meshes numbered in mesh order (a device scan over the template's meshes; after k_tmpl_meshes): tmesh[m].pad[1] = number + 1,
// trmesh[number] = (mesh, its first element among the instance's Round-join elements); trmesh[count] = (~0, the total); the count ->
// cls[nclasses].pad[1]. Only launched for templates that hold Round joins (k_tmpl_meshes leaves pad[1] = 0).
struct OpTmplRoundIndex
{
	VgxTmplBuild B;
	__device__ uint64_t size() const { return B.num_meshes; }
	__device__ Sum3 load(uint64_t m) const
	{
		Sum3 r = sum3_zero();
		const VgxMeshDesc md = B.mdesc[m];
		if (tmpl_is_round(md.kind)) { r.a = 1; r.b = md.poly_n; }
		return r;
	}
	__device__ void store(uint64_t m, Sum3 e) const
	{
		const VgxMeshDesc md = B.mdesc[m];
		if (!tmpl_is_round(md.kind)) { return; }
		B.tmesh[m].pad[1] = (uint32_t)e.a + 1u;
		B.tmsz[m] = make_uint2(0x80000000u | (uint32_t)e.a, 0u);
		VgxTmplRoundMesh r; r.mesh = (uint32_t)m; r.elem0 = (uint32_t)e.b;
		B.trmesh[e.a] = r;
		// what the emit kernel needs of a Round-join mesh besides its record, in the record (l2: only AA fills read the three local vertices):
		// the arc step da (stroker.cpp:1398 -- scale, half width and tolerance are the template's) and the mesh's first table word
		const vgx_draw* td = B.draws + md.draw;
		B.tmesh[m].l2[0] = vgx_step_angle(td->scale, B.tmesh[m].f0, td->tess_tol);
		B.tmesh[m].l2[1] = __uint_as_float(r.elem0);
	}
	__device__ void finish(Sum3 tot) const
	{
		VgxTmplRoundMesh r; r.mesh = ~0u; r.elem0 = (uint32_t)tot.b;
		B.trmesh[tot.a] = r;
		B.cls[B.nclasses].pad[1] = (uint32_t)tot.a;
	}
};

// Several classes: where each class's Round-join meshes begin in trmesh (they are numbered across the concatenated template) -> cls[c].pad[1],
// and the element number of its first one -> cls[c].pad[0], c < nclasses (entry [nclasses] keeps the totals / the style bits). One thread.
__global__ void k_tmpl_round_classes(VgxTmplBuild B)
{
	const uint32_t R = B.cls[B.nclasses].pad[1];
	for (uint32_t c = 0; c < B.nclasses; ++c) {
		uint32_t lo = 0, hi = R; // first Round-join mesh at or behind the class's first mesh
		while (lo < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			if (B.trmesh[mid].mesh < B.cls[c].mesh0) { lo = mid + 1; } else { hi = mid; }
		}
		B.cls[c].pad[1] = lo;
		B.cls[c].pad[0] = B.trmesh[lo].elem0; // (trmesh[R] = the total)
	}
}

// Element table in processing order: tiles of `tile` elements of the instance's output-ordered element stream; inside a
// tile the fill elements first, then the stroke elements (both in output order). Every class starts a tile of its own
// (tile cls.tile0, table slot cls.tile0 * tile): a tile never holds elements of two classes.
#define VGX_TMPL_ELEMS_MAXM 1032 /* meshes of one tile k_tmpl_elems keeps in LDS (a mesh has >= 2 elements: a 2 048-element tile holds <= 1 025) */
__global__ __launch_bounds__(256) void k_tmpl_elems(VgxTmplBuild B) // one workgroup per tile
{
	const uint64_t M = B.num_meshes;
	const uint64_t E = B.num_elems;
	const uint64_t T = B.tile;
	const uint32_t tid = threadIdx.x;
	// first output-ordered element of mesh m (= elements in front of it); zero-length entries cannot occur (every mesh has >= 2 elements)
	auto first = [&](uint64_t m) { return B.prefix_fill[m] + B.prefix_stroke[m]; };
	__shared__ unsigned long long s_pf[VGX_TMPL_ELEMS_MAXM], s_ps[VGX_TMPL_ELEMS_MAXM]; // the tile's meshes: fill / stroke elements in front (bit 63 of s_ps: a fill mesh)
	__shared__ unsigned long long s_poly[VGX_TMPL_ELEMS_MAXM];                          // ... first vertex of the mesh's polyline
	__shared__ uint32_t s_pick;
	__shared__ uint64_t s_b[6]; // m0, f0, s0, m1, f1, (unused)
	// the last mesh in [lo, hi) whose first element is <= x (first(lo) <= x): the whole workgroup searches, 256 probes per round trip -- one
	// thread walking a binary search over the millions of meshes of a static batch was 22 dependent round trips per tile, with 255 threads
	// waiting (round 6: k_tmpl_elems 2.0 ms of a 7.4 ms count)
	auto lastLe = [&](uint64_t x, uint64_t lo, uint64_t hi) {
		while (hi - lo > 1) { // workgroup-uniform
			const uint64_t step = (hi - lo + 255) / 256;
			const uint64_t m = lo + (uint64_t)tid * step;
			if (tid == 0) { s_pick = 0; }
			__syncthreads();
			if (tid > 0 && m < hi && first(m) <= x) { atomicMax(&s_pick, tid); }
			__syncthreads();
			const uint32_t k = s_pick;
			__syncthreads();
			lo += (uint64_t)k * step;
			hi = lo + step < hi ? lo + step : hi;
		}
		return lo;
	};
	for (uint64_t tile = blockIdx.x; tile < B.cls[B.nclasses].tile0; tile += gridDim.x) {
		uint32_t c = 0;
		while (c + 1 < B.nclasses && B.cls[c + 1].tile0 <= tile) { ++c; }
		const VgxTmplClass cl = B.cls[c];
		const uint64_t cEnd = B.cls[c + 1].elem0;
		const uint64_t x0 = cl.elem0 + (tile - cl.tile0) * T; // the tile's first element (output order of the concatenated template)
		const uint64_t x1 = x0 + T < cEnd ? x0 + T : cEnd;
		// the tile's bounds: the mesh that owns its first element, and the one that owns the element behind its last (M: none)
		const uint64_t m0 = lastLe(x0, 0, M);
		const uint64_t m1 = x1 >= E ? M : lastLe(x1, m0, (m0 + T / 2 + 2 < M) ? m0 + T / 2 + 2 : M);
		const uint64_t hi = m1 < M ? m1 + 1 : M;
		const bool inLds = hi - m0 <= (uint64_t)VGX_TMPL_ELEMS_MAXM;
		__syncthreads();
		if (inLds) {
			for (uint64_t m = m0 + tid; m < hi; m += 256) {
				const VgxMeshDesc md = B.mdesc[m];
				s_pf[m - m0] = B.prefix_fill[m];
				s_ps[m - m0] = B.prefix_stroke[m] | (VGX_MD_KIND(md.kind) < VGX_MESH_STROKE ? (1ull << 63) : 0ull);
				s_poly[m - m0] = md.poly_first;
			}
		}
		if (tid == 0) {
			// fill / stroke elements in front of element x that mesh m owns (m = M: the totals)
			auto before = [&](uint64_t x, uint64_t m, uint64_t* fB, uint64_t* sB, uint32_t* j) {
				if (m >= M) { *fB = B.prefix_fill[M]; *sB = B.prefix_stroke[M]; *j = 0; return; }
				const uint64_t pf = B.prefix_fill[m], psk = B.prefix_stroke[m];
				const uint32_t jj = (uint32_t)(x - (pf + psk));
				const bool f = VGX_MD_KIND(B.mdesc[m].kind) < VGX_MESH_STROKE;
				*j = jj; *fB = pf + (f ? jj : 0u); *sB = psk + (f ? 0u : jj);
			};
			uint64_t f0, s0, f1, s1;
			uint32_t j0, j1;
			before(x0, m0, &f0, &s0, &j0);
			before(x1, x1 >= E ? M : m1, &f1, &s1, &j1);
			s_b[0] = m0; s_b[1] = f0; s_b[2] = s0; s_b[3] = m1; s_b[4] = f1;
			// tile record, first half: the mesh that owns the tile's first element; bit 31: it begins exactly here
			B.ttile[tile].mesh0 = (uint32_t)m0 | (j0 == 0 ? 0x80000000u : 0u);
			B.ttile[tile].nel = (uint32_t)(x1 - x0);
		}
		__syncthreads();
		const uint64_t f0 = s_b[1], s0 = s_b[2], f1 = s_b[4];
		for (uint64_t e = x0 + tid; e < x1; e += blockDim.x) {
			// owner of element e among the tile's meshes [m0, hi): the last one whose first element is <= e
			uint64_t lo = m0, up = hi, pf, psk, polyFirst;
			bool isFill;
			if (inLds) {
				while (up - lo > 1) {
					const uint64_t mid = (lo + up) >> 1;
					if (s_pf[mid - m0] + (s_ps[mid - m0] & ~(1ull << 63)) <= e) { lo = mid; } else { up = mid; }
				}
				pf = s_pf[lo - m0]; psk = s_ps[lo - m0] & ~(1ull << 63); isFill = (s_ps[lo - m0] >> 63) != 0; polyFirst = s_poly[lo - m0];
			} else { // (tiles of a tile size above 2 048: the same search in memory)
				while (up - lo > 1) {
					const uint64_t mid = (lo + up) >> 1;
					if (first(mid) <= e) { lo = mid; } else { up = mid; }
				}
				pf = B.prefix_fill[lo]; psk = B.prefix_stroke[lo]; isFill = VGX_MD_KIND(B.mdesc[lo].kind) < VGX_MESH_STROKE; polyFirst = B.mdesc[lo].poly_first;
			}
			const uint64_t m = lo;
			const uint32_t j = (uint32_t)(e - (pf + psk));
			const uint64_t f = pf + (isFill ? j : 0u), sk = psk + (isFill ? 0u : j);
			// inside a tile the fill elements first, then the stroke elements (both in output order)
			const uint64_t slot = tile * T + (isFill ? f - f0 : (f1 - f0) + (sk - s0));
			VgxTmplElem r;
			r.mesh = (uint32_t)m;
			r.jq = j | ((uint32_t)(e - x0) << 16); // j < 65536 (a mesh holds at most 65536 vertices), position in the tile < tile <= 65536
			const float2 lv = B.poly[polyFirst + j];
			r.lx = lv.x; r.ly = lv.y;
			B.telem[slot] = r;
		}
		__syncthreads(); // the LDS tables are the next tile's
	}
}

// Tile records, second half (after k_tmpl_elems): the draws of the period whose records the tile needs = the draws of its
// meshes; the first / last tile also take the draws in front of / behind every mesh, so that the tiles together cover the period
// (every draw record is verified by some workgroup of every instance).
__global__ __launch_bounds__(256) void k_tmpl_tiles(VgxTmplBuild B)
{
	const uint32_t nt = B.cls[B.nclasses].tile0;
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nt) { return; }
	uint32_t c = 0;
	while (c + 1 < B.nclasses && B.cls[c + 1].tile0 <= t) { ++c; }
	const bool first = t == B.cls[c].tile0, last = t + 1 == B.cls[c + 1].tile0; // of its class
	const uint32_t mA = B.ttile[t].mesh0 & 0x7FFFFFFFu;
	uint32_t mB = B.cls[c + 1].mesh0 - 1;
	if (!last) { const uint32_t nx = B.ttile[t + 1].mesh0; mB = (nx & 0x7FFFFFFFu) - (nx >> 31); }
	const uint32_t d0 = c * B.period;
	// A tile whose first mesh begins exactly here also takes the mesh-less draws in front of it (behind the previous tile's last
	// mesh): the tiles' draw ranges partition [0, period), so a change to ANY draw record is seen by some workgroup.
	uint32_t dA = first ? 0u : B.mdesc[mA].draw - d0;
	if (!first && (B.ttile[t].mesh0 >> 31) != 0 && mA > B.cls[c].mesh0) {
		const uint32_t prev = B.mdesc[mA - 1].draw - d0 + 1;
		dA = prev < dA ? prev : dA;
	}
	const uint32_t dB = last ? B.period - 1 : B.mdesc[mB].draw - d0;
	B.ttile[t].mesh_last = mB;
	B.ttile[t].draw0 = dA;
	B.ttile[t].ndraws = dB - dA + 1;
	B.ttile[t].cmesh0 = B.cls[c].mesh0;
	B.ttile[t].cdraw0 = d0;
	B.ttile[t].pad = 0;
}

// ---- step ------------------------------------------------------------------------------------------------------------
// One workgroup = one TILE of one instance: `tile` consecutive elements of the instance's output-ordered element stream, i.e.
// a contiguous piece of each output stream (~40 KB), a contiguous range of the template's meshes and of the period's draws.
// What bounds the kernel is the latency of its dependent loads (measured with -DVGX_TMPL_PROFILE: a workgroup that fetched
// tile -> mesh records -> draw records -> vertices one after the other spent 60 % of its life waiting for them), so everything
// is addressed from the tile record and requested at once:
//   phase 0  a) one thread per draw of the tile: the instance's draw record (HBM) -- verified against the saved first period,
//               transform + colours parked in LDS;  one thread per mesh: the template's mesh record and, for AA fills, the
//               polygon's first three vertices;  every thread: its elements' records and template vertices (phase 1's loads)
//            b) one thread per mesh: per-mesh record in LDS -- colour, widths, output offsets, and for AA fills the orientation
//               of the TRANSFORMED polygon's first triangle (stroker.cpp:721-723), once per mesh instead of once per corner;
//               the caller's mesh table for the meshes that begin in this tile
//   phase 1  one lane per element: transformPos2D (vg_util.h:24-28) of its vertex ONCE, parked in LDS at the element's
//            output-order position inside the tile ("the growing polyline staged in LDS")
//   phase 2  one lane per element: vec2Dir(own vertex, next vertex) (stroker.cpp:31-38) once, parked in LDS
//   phase 3  one lane per element: the edge directions around it from LDS (from L2 + transform for the handful of elements
//            whose neighbour lies in another tile), calcExtrusionVector and the rest of the stroker's per-element arithmetic,
//            stores at closed-form addresses (workgroup-uniform stream bases + 32-bit offsets).
struct TmplXf { float m0, m1, m2, m3, m4, m5; };
__device__ __forceinline__ V2 tmpl_xf(const TmplXf& m, float2 p) // transformPos2D, vg_util.h:24-28
{
	return v2(m.m0 * p.x + m.m2 * p.y + m.m4, m.m1 * p.x + m.m3 * p.y + m.m5);
}

struct __attribute__((aligned(16))) TmplDraw // per draw of the tile, in LDS. 32 bytes
{
	float m0, m1, m2, m3;
	float m4, m5; uint32_t fill_color, stroke_color;
};
struct __attribute__((aligned(16))) TmplRec // per mesh of the tile, in LDS. 32 bytes
{
	uint32_t ibase, n, v_off, i_off;    // ibase: assembly armed: vertices in front of the mesh inside its draw command (added to every index); else 0
	uint32_t kind, color; float f0, f1; // kind: VgxMeshDesc::kind word | (the mesh's draw, relative to the tile's first draw) << 16
	                                    // fills: f0 = aa WITH the instance's orientation sign; strokes: hsw, hswAA
};
#define TMPL_REC_DK(r) ((r)->kind >> 16)
__device__ __forceinline__ TmplXf tmpl_draw_xf(const TmplDraw* d)
{
	TmplXf xf; xf.m0 = d->m0; xf.m1 = d->m1; xf.m2 = d->m2; xf.m3 = d->m3; xf.m4 = d->m4; xf.m5 = d->m5;
	return xf;
}

// One draw record of the instance: checks (the ordinary path's finiteness checks + equality with the saved first period in
// every field the template depends on) and the part the emit needs.
__device__ __forceinline__ TmplDraw tmpl_load_draw(const VgxTmplArgs& A, const vgx_draw* idraws, const vgx_draw* tdraws, uint32_t k)
{
	const uint4* q = (const uint4*)(idraws + k);
	const uint4* t = (const uint4*)(tdraws + k);
	const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
	const uint4 t0 = t[0], t1 = t[1], t2 = t[2];
	const uint32_t e = tmpl_validate(q0, q1, q2, q3, A.npaths);
	if (e != VGX_OK) { set_status(A.totals, e); }
	else if (!tmpl_same(q0, q1, q2, t0, t1, t2)) { set_status(A.totals, VGX_E_STALE); }
	TmplDraw d;
	d.m0 = __uint_as_float(q2.y); d.m1 = __uint_as_float(q2.z); d.m2 = __uint_as_float(q2.w);
	d.m3 = __uint_as_float(q3.x); d.m4 = __uint_as_float(q3.y); d.m5 = __uint_as_float(q3.z);
	d.fill_color = q0.z; d.stroke_color = q1.x;
	return d;
}

// orientation of the first triangle of the TRANSFORMED polygon (stroker.cpp:721-723) -> aa with its sign
__device__ __forceinline__ float tmpl_fill_aa(const TmplXf& xf, float2 l0, float2 l1, float2 l2, float halfFringe)
{
	const V2 a0 = tmpl_xf(xf, l0), a1 = tmpl_xf(xf, l1), a2 = tmpl_xf(xf, l2);
	const float orient = v2cross(v2sub(a1, a0), v2sub(a2, a0));
	return halfFringe * vgm_sign(orient);
}

// One element of an OPEN stroke with MITER joins and Butt / Square caps, AA (4 rails) or Thin (3 rails; its caps are Butt or
// "everything else", stroker.cpp:2012-2058): the commonest open style, fixed sizes like the closed one -- R vertices per element,
// AA caps two triangles of their own (stroker.cpp:1422-1474, 1858-1916), a bridge from every element to the next. Same arithmetic
// as elem_emit (vgx_elem.h) for these cases, without its register stage; only the general kernel instantiation contains it.
__device__ __forceinline__ void tmpl_stroke_elem_open(const TmplOut& O, uint32_t kindWord, uint32_t N, uint32_t vOff, uint32_t iOff, uint32_t ibase, uint32_t color, float hsw, float hswAA,
	const VgxTmplMesh* tmm, uint32_t j, V2 p1, V2 dPrev2, V2 dPrev, V2 d12)
{
	const bool thin = VGX_MD_KIND(kindWord) == VGX_MESH_STROKE_AA_THIN;
	const uint32_t cap = VGX_MD_CAP(kindWord);
	const uint32_t R = thin ? 3u : 4u;
	const uint32_t bridgeIdx = thin ? 12u : 18u;
	const uint32_t capNi = thin ? 0u : 6u;
	const float sideWidth = thin ? hsw : hswAA; // fringe : hswAA
	const bool first = j == 0, last = j + 1 == N;
	const uint32_t b = R * j;
	const uint32_t bi = b + ibase, top = bi + R - 1;
	const uint32_t c0 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
	bool L = true; // caps connect like a join whose left side is the inner one (rails b .. b + R - 1 in order)
	V2 q0, q1, q2, q3;
	if (first || last) {
		const V2 d = first ? d12 : dPrev;
		const V2 l = v2ccw(d);
		if (thin) { // :2012-2058, 2242-2294
			const V2 lf = v2mul(l, hsw);
			q1 = p1;
			if (cap == VGX_CAP_BUTT) { q0 = v2add(p1, lf); q2 = v2sub(p1, lf); }
			else {
				const V2 df = v2mul(d, hsw);
				q0 = first ? v2add(p1, v2sub(lf, df)) : v2add(p1, v2add(lf, df));
				q2 = first ? v2sub(p1, v2add(lf, df)) : v2sub(p1, v2sub(lf, df));
			}
			q3 = q2;
		} else {
			const V2 lh = v2mul(l, hsw);
			const V2 lhaa = v2mul(l, hswAA);
			if (cap == VGX_CAP_BUTT) { // :1422-1447, 1858-1886
				const V2 daa = v2mul(d, __uint_as_float(tmm->pad[0])); // the draw's fringe
				q0 = first ? v2add(p1, v2sub(lhaa, daa)) : v2add(p1, v2add(lhaa, daa));
				q1 = v2add(p1, lh);
				q2 = v2sub(p1, lh);
				q3 = first ? v2sub(p1, v2add(lhaa, daa)) : v2sub(p1, v2sub(lhaa, daa));
			} else { // Square, :1448-1474, 1887-1916
				const V2 dh = v2mul(d, hsw);
				const V2 dhaa = v2mul(d, hswAA);
				q0 = first ? v2add(p1, v2sub(lhaa, dhaa)) : v2add(p1, v2add(lhaa, dhaa));
				q1 = first ? v2add(p1, v2sub(lh, dh)) : v2add(p1, v2add(lh, dh));
				q2 = first ? v2sub(p1, v2add(lh, dh)) : v2sub(p1, v2sub(lh, dh));
				q3 = first ? v2sub(p1, v2add(lhaa, dhaa)) : v2sub(p1, v2sub(lhaa, dhaa));
			}
		}
	} else {
		const VgxJoin jn = vgx_join_dirs(dPrev, d12, sideWidth);
		L = jn.leftInner;
		if (thin) { // :2060-2110
			const V2 vf = v2mul(jn.v, hsw);
			q0 = L ? v2add(p1, vf) : v2sub(p1, vf);
			q1 = p1;
			q2 = L ? v2sub(p1, vf) : v2add(p1, vf);
			q3 = q2;
		} else { // :1524-1579
			const V2 vhaa = v2mul(jn.v, hswAA);
			const V2 vh = v2mul(jn.v, hsw);
			q0 = L ? v2add(p1, vhaa) : v2sub(p1, vhaa);
			q1 = L ? v2add(p1, vh) : v2sub(p1, vh);
			q2 = L ? v2sub(p1, vh) : v2add(p1, vh);
			q3 = L ? v2sub(p1, vhaa) : v2add(p1, vhaa);
		}
	}
	const Rails mine = thin ? (L ? rails(bi, bi + 1, bi + 2, 0) : rails(top, bi + 1, bi, 0)) : (L ? rails(bi, bi + 1, bi + 2, bi + 3) : rails(top, bi + 2, bi + 1, bi));
	char* pp = O.pos + (vOff + b) * 8u;
	char* pc = O.col + (vOff + b) * 4u;
	if (thin) {
		PosPair q; q.x0 = q0.x; q.y0 = q0.y; q.x1 = q1.x; q.y1 = q1.y;
		ColPair c; c.c0 = c0; c.c1 = color;
		*(PosPair*)pp = q;
		*(float2*)(pp + 16) = make_float2(q2.x, q2.y);
		*(ColPair*)pc = c;
		*(uint32_t*)(pc + 8) = c0;
	} else {
		PosPair q; q.x0 = q0.x; q.y0 = q0.y; q.x1 = q1.x; q.y1 = q1.y;
		PosPair r; r.x0 = q2.x; r.y0 = q2.y; r.x1 = q3.x; r.y1 = q3.y;
		ColPair c; c.c0 = c0; c.c1 = color;
		ColPair d; d.c0 = color; d.c1 = c0;
		*(PosPair*)pp = q;
		*(PosPair*)(pp + 16) = r;
		*(ColPair*)pc = c;
		*(ColPair*)(pc + 8) = d;
	}
	if (first && !thin) { // the cap's quad (0, 2, 1) (0, 3, 2), :1443-1446
		Idx6 t; t.a = (bi & 0xFFFFu) | ((bi + 2) << 16); t.b = ((bi + 1) & 0xFFFFu) | (bi << 16); t.c = ((bi + 3) & 0xFFFFu) | ((bi + 2) << 16);
		*(Idx6*)(O.idx + iOff * 2u) = t;
	}
	if (j > 0) { // the bridge from the previous element (stroker.cpp:1557-1564, 1876-1883; thin :2093-2098, 2262-2267)
		bool pL = true;
		if (j > 1) { pL = vgx_join_dirs(dPrev2, dPrev, sideWidth).leftInner; }
		const uint32_t pb = R * (j - 1) + ibase, ptop = pb + R - 1;
		const Rails p = thin ? (pL ? rails(pb, pb + 1, pb + 2, 0) : rails(ptop, pb + 1, pb, 0))
		                     : (pL ? rails(pb, pb + 1, pb + 2, pb + 3) : rails(ptop, pb + 2, pb + 1, pb));
		char* pi = O.idx + (iOff + capNi + bridgeIdx * (j - 1)) * 2u;
		Idx6 t0; t0.a = (p.a & 0xFFFFu) | (p.b << 16); t0.b = (mine.b & 0xFFFFu) | (p.a << 16); t0.c = (mine.b & 0xFFFFu) | (mine.a << 16);
		Idx6 t1; t1.a = (p.b & 0xFFFFu) | (p.c << 16); t1.b = (mine.c & 0xFFFFu) | (p.b << 16); t1.c = (mine.c & 0xFFFFu) | (mine.b << 16);
		*(Idx6*)pi = t0;
		*(Idx6*)(pi + 12) = t1;
		if (!thin) {
			Idx6 t2; t2.a = (p.c & 0xFFFFu) | (p.d << 16); t2.b = (mine.d & 0xFFFFu) | (p.c << 16); t2.c = (mine.d & 0xFFFFu) | (mine.c << 16);
			*(Idx6*)(pi + 24) = t2;
			if (last) { // the end cap's quad (b, b + 1, b + 2) (b, b + 2, b + 3), :1884-1885
				Idx6 t; t.a = (bi & 0xFFFFu) | ((bi + 1) << 16); t.b = ((bi + 2) & 0xFFFFu) | (bi << 16); t.c = ((bi + 2) & 0xFFFFu) | ((bi + 3) << 16);
				*(Idx6*)(pi + 36) = t;
			}
		}
	}
}
__device__ __forceinline__ bool tmpl_stroke_is_open_fast(uint32_t kindWord)
{
	const uint32_t kind = VGX_MD_KIND(kindWord);
	return VGX_MD_CLOSED(kindWord) == 0 && VGX_MD_JOIN(kindWord) == VGX_JOIN_MITER
		&& (kind == VGX_MESH_STROKE_AA_THIN || (kind == VGX_MESH_STROKE_AA && VGX_MD_CAP(kindWord) != VGX_CAP_ROUND));
}

// Where one instance lies in the batch (workgroup-uniform): output bases, first mesh, first draw; its class's saved draw records
// and first template mesh.
struct TmplPlace
{
	uint64_t v, i;     // first output vertex / index
	uint64_t m;        // first mesh in the batch's mesh sequence
	uint32_t draw0;    // first draw (= instance * period)
	uint32_t cmesh0;   // first template mesh of the class
	const vgx_draw* tdraws; // the class representative's draw records
};
// the caller's mesh table: the template's record moved to this instance
__device__ __forceinline__ void tmpl_mesh_out(const VgxTmplArgs& A, const TmplPlace& P, uint32_t mesh)
{
	vgx_mesh mr = A.tmtab[mesh];
	mr.first_vertex += P.v;
	mr.first_index += P.i;
	mr.draw += P.draw0;
	A.meshes_out[P.m + (mesh - P.cmesh0)] = mr;
}

// Round joins: the places and sizes are the instance's (per-step table)
__device__ __forceinline__ void tmpl_mesh_out_placed(const VgxTmplArgs& A, const TmplPlace& P, uint32_t mesh, uint4 mi)
{
	vgx_mesh mr = A.tmtab[mesh];
	mr.first_vertex = P.v + mi.x;
	mr.first_index = P.i + mi.y;
	mr.num_vertices = mi.z;
	mr.num_indices = mi.w;
	mr.draw += P.draw0;
	A.meshes_out[P.m + (mesh - P.cmesh0)] = mr;
}

// ---- every other stroke whose SIZES do not depend on the geometry: open strokes with Butt / Square / Round caps, Bevel joins,
// non-AA strokes (only Round JOINS count their points on the transformed polyline, stroker.cpp:1146, 1592). The general element
// code of vgx_elem.h (elem_geometry / elem_emit, what k_stroke runs) on the staged vertices, with what the sequential stroker
// carries from element to element in closed form: the element's first vertex / index inside its mesh, and the previous
// element's exit rails recomputed from ITS geometry (same inputs, same bits).
struct TmplVtx01 // elem_emit reads vertices 0 and 1 only (closing bridge); scalar members + selects: a `cond ? v0 : v1` of two V2 members became an
{                // indexed load and put the whole mesh context into scratch memory (1.9 GB read + 4.9 GB written per step on the Tiger with Bevel joins)
	float x0, y0, x1, y1;
	__device__ __forceinline__ V2 ld(uint32_t i) const { const bool z = i == 0; return v2(z ? x0 : x1, z ? y0 : y1); }
};

// vertices / own indices of a cap and of a join of this stroke flavour (elem_geometry's counts, vgx_elem.h), H = numPointsHalfCircle
__device__ __forceinline__ void tmpl_stroke_counts(uint32_t kind, uint32_t cap, uint32_t join, uint32_t H, uint32_t* capNv, uint32_t* capNiFirst, uint32_t* joinNv, uint32_t* joinNi, uint32_t* bridge)
{
	const uint32_t R = (kind == VGX_MESH_STROKE) ? 2u : (kind == VGX_MESH_STROKE_AA ? 4u : 3u);
	*bridge = (R - 1) * 6;
	const bool roundCap = cap == VGX_CAP_ROUND && kind != VGX_MESH_STROKE_AA_THIN;
	if (roundCap) {
		*capNv = kind == VGX_MESH_STROKE_AA ? 2 * H : H;
		*capNiFirst = kind == VGX_MESH_STROKE_AA ? 9 * H - 12 : 3 * (H - 2);
	} else {
		*capNv = R;
		*capNiFirst = kind == VGX_MESH_STROKE_AA ? 6u : 0u;
	}
	if (kind == VGX_MESH_STROKE_AA_THIN) { const bool bevel = join != VGX_JOIN_MITER; *joinNv = bevel ? 4u : 3u; *joinNi = bevel ? 3u : 0u; }
	else if (join == VGX_JOIN_MITER) { *joinNv = R; *joinNi = 0; }
	else if (kind == VGX_MESH_STROKE_AA) { *joinNv = 6; *joinNi = 9; } // Bevel = an arc of one segment: 2n + 4, 9n (stroker.cpp:1599, 1675)
	else { *joinNv = 3; *joinNi = 3; }                                  // n + 2, 3n (:1156, 1186)
}

// Inlined ONCE per kernel (the tile loop calls it from a rolled loop, see tmpl_elem_emit's PASS): inlined four times into the
// unrolled element loop it spilled, as a real function its callee-saved registers went through scratch on every call. Everything
// it needs comes by value: own vertex, previous vertex, the three edge directions around the element, the mesh's first two
// vertices (closing bridge).
__device__ __forceinline__ void tmpl_stroke_general(char* opos, char* ocol, char* oidx, uint32_t kindWord, uint32_t N, uint32_t vOff, uint32_t iOff, uint32_t ibase, uint32_t color,
	float hsw, float hswAA, float fringe, const vgx_draw* tdraw, uint32_t j, V2 p1, V2 pPrev, V2 d12, V2 dPrev, V2 dPrev2, V2 v0, V2 v1, bool placed, uint32_t bPlaced, uint32_t kPlaced, uint32_t nvPrevPlaced, bool prevInner, float daPre)
{
	MeshCtxT<TmplVtx01> mc;
	mc.da = daPre;
	mc.kind = VGX_MD_KIND(kindWord); mc.closed = VGX_MD_CLOSED(kindWord) != 0; mc.cap = VGX_MD_CAP(kindWord); mc.join = VGX_MD_JOIN(kindWord);
	mc.N = N; mc.j = j; mc.hsw = hsw; mc.hswAA = hswAA; mc.fringe = fringe; mc.dr = tdraw; mc.vtx.x0 = v0.x; mc.vtx.y0 = v0.y; mc.vtx.x1 = v1.x; mc.vtx.y1 = v1.y;
	uint32_t H = 2;
	if (!mc.closed && mc.cap == VGX_CAP_ROUND && mc.kind != VGX_MESH_STROKE_AA_THIN) { H = vgx_half_circle_points(mesh_da(mc)); }
	uint32_t capNv, capNi, joinNv, joinNi, bridge;
	tmpl_stroke_counts(mc.kind, mc.cap, mc.join, H, &capNv, &capNi, &joinNv, &joinNi, &bridge);
	// first vertex / index of element jj inside the mesh
	auto vbase = [&](uint32_t jj) { return mc.closed ? jj * joinNv : (jj == 0 ? 0u : capNv + (jj - 1) * joinNv); };
	auto ibaseOf = [&](uint32_t jj) { return jj == 0 ? 0u : (mc.closed ? joinNi : capNi) + (jj - 1) * (bridge + joinNi); };
	const Elem e = elem_geometry(mc, p1, dPrev, d12);
	Rails prev = rails(0, 0, 0, 0);
#ifndef VGX_TMPL_ROUND_PREV_TABLE
#define VGX_TMPL_ROUND_PREV_TABLE 1 /* 0 (measurement): the previous element's geometry evaluated again, its place = own place - its vertices */
#endif
	if (VGX_TMPL_ROUND_PREV_TABLE && e.hasConnect && placed) {
		// Round-join meshes: the previous element's place and inner side come from the per-step table (k_tmpl_round_sizes evaluated its
		// geometry already), its arc's point count from its size -- what its exit rails are made of (elem_exit_rails)
		const uint32_t nvPrev = nvPrevPlaced, bPrev = bPlaced - nvPrev;
		Elem ep = e;
		ep.et = (!mc.closed && j == 1) ? ET_CAP_FIRST : ET_JOIN;
		ep.leftInner = prevInner;
		ep.arc.n = mc.kind == VGX_MESH_STROKE_AA ? (nvPrev - 4u) >> 1 : nvPrev - 2u; // 2n + 4 / n + 2 vertices per join (stroker.cpp:1599, 1156)
		ep.H = H;
		prev = elem_exit_rails(mc, ep, bPrev);
	} else if (e.hasConnect) { // the previous element's exit rails (prevSegment*ID, stroker.cpp:1401-1410), from its own geometry
		mc.j = j - 1;
		const Elem ep = elem_geometry(mc, pPrev, dPrev2, dPrev);
		prev = elem_exit_rails(mc, ep, placed ? bPlaced - ep.nv : vbase(j - 1));
		mc.j = j;
	}
	StrokeWriter w;
	w.pos = (float*)(opos + (size_t)vOff * 8u); w.col = (uint32_t*)(ocol + (size_t)vOff * 4u); w.idx = (uint16_t*)(oidx + (size_t)iOff * 2u);
	w.color = color; w.c0 = color & 0x00FFFFFFu; w.ib = ibase;
	w.reset();
	const uint32_t b = placed ? bPlaced : vbase(j), k = placed ? kPlaced : ibaseOf(j);
	elem_emit(mc, e, b, k, prev, w);
	w.flush(b, k);
}

// Closed strokes with Bevel joins, AA (4 rails; stroker.cpp:1580-1691 with numArcPoints = 1, closing bridge :1970-1984) or Thin (3 rails;
// :2112-2180, closing :2295-2306; the thin stroker turns Round joins into Bevel ones as well, :318-327): nothing is data dependent, so -- like
// tmpl_stroke_elem for Miter joins -- every element writes its own join (6 / 4 vertices, one / three fan triangles of the bevel) and the bridge
// that ENDS at it (join 0: the closing bridge, at the end of the mesh's index range), with the previous join's inner side recomputed from the
// staged directions. Index layout of the mesh (elem_emit's): join 0's own triangles, then per join j >= 1 its bridge and its own triangles,
// then the closing bridge. Same values at the same places as the general body (tests: VGX_TMPL=0 byte for byte), a third of its instructions.
__device__ __forceinline__ bool tmpl_stroke_is_closed_bevel(uint32_t kindWord)
{
	const uint32_t kind = VGX_MD_KIND(kindWord), join = VGX_MD_JOIN(kindWord);
	return VGX_MD_CLOSED(kindWord) != 0 && ((kind == VGX_MESH_STROKE_AA && join == VGX_JOIN_BEVEL) || (kind == VGX_MESH_STROKE_AA_THIN && join != VGX_JOIN_MITER));
}
__device__ __forceinline__ void tmpl_stroke_elem_bevel(const TmplOut& O, uint32_t kindWord, uint32_t N, uint32_t vOff, uint32_t iOff, uint32_t ibase, uint32_t color, float hsw, float hswAA,
	float fringe, uint32_t j, V2 p1, V2 dPrev2, V2 dPrev, V2 d12)
{
	const bool thin = VGX_MD_KIND(kindWord) == VGX_MESH_STROKE_AA_THIN;
	const uint32_t R = thin ? 4u : 6u;             // vertices per join
	const uint32_t ownIdx = thin ? 3u : 9u;        // the bevel's own triangles
	const uint32_t bridgeIdx = thin ? 12u : 18u;
	const float sideWidth = thin ? fringe : hswAA; // elem_geometry
	const VgxJoin jn = vgx_join_dirs(dPrev, d12, sideWidth);
	const bool L = jn.leftInner;
	const uint32_t b = R * j;
	const uint32_t bi = b + ibase; // index VALUES carry the assembly base, positions in the streams do not
	const uint32_t c0 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
	const V2 n01 = L ? v2cw(jn.d01) : v2ccw(jn.d01);
	const V2 n12 = L ? v2cw(jn.d12) : v2ccw(jn.d12);
	char* pp = O.pos + (vOff + b) * 8u;
	char* pc = O.col + (vOff + b) * 4u;
	const uint32_t kOwn = j == 0 ? 0u : ownIdx + (j - 1u) * (bridgeIdx + ownIdx) + bridgeIdx; // behind the bridge that ends here
	char* pio = O.idx + (iOff + kOwn) * 2u;
	Rails mine; // entry rails
	if (thin) {
		const V2 vf = v2mul(jn.v, fringe);
		const V2 q0 = L ? v2add(p1, vf) : v2sub(p1, vf);
		const V2 q2 = v2add(p1, v2mul(n01, fringe));
		const V2 q3 = v2add(p1, v2mul(n12, fringe));
		PosPair q; q.x0 = q0.x; q.y0 = q0.y; q.x1 = p1.x; q.y1 = p1.y;
		PosPair r; r.x0 = q2.x; r.y0 = q2.y; r.x1 = q3.x; r.y1 = q3.y;
		ColPair c; c.c0 = c0; c.c1 = color;
		ColPair d; d.c0 = c0; d.c1 = c0;
		Idx3 t; // the bevel: (b+1, b+2, b+3) / (b+1, b+3, b+2)
		t.a = ((bi + 1u) & 0xFFFFu) | ((L ? bi + 2u : bi + 3u) << 16); t.b = (uint16_t)(L ? bi + 3u : bi + 2u);
		VGX_ST_GUARD(c0 ^ __float_as_uint(q.x0) ^ __float_as_uint(r.y1)) {
		*(PosPair*)pp = q;
		*(PosPair*)(pp + 16) = r;
		*(ColPair*)pc = c;
		*(ColPair*)(pc + 8) = d;
		TMPL_IDX_ON *(Idx3*)pio = t;
		}
		mine = L ? rails(bi, bi + 1, bi + 2, 0) : rails(bi + 2, bi + 1, bi, 0);
	} else {
		const V2 vhaa = v2mul(jn.v, hswAA);
		const V2 vh = v2mul(jn.v, hsw);
		const V2 q0 = L ? v2add(p1, vhaa) : v2sub(p1, vhaa);
		const V2 q1 = L ? v2add(p1, vh) : v2sub(p1, vh);
		const float cosAngle = vgm_abs(v2dot(n01, n12));
		const V2 q2 = v2sub(v2add(p1, v2mul(n01, hsw)), v2mul(jn.d01, cosAngle * fringe));
		const V2 q3 = v2add(p1, v2mul(n01, hswAA));
		const V2 q4 = v2add(v2add(p1, v2mul(n12, hsw)), v2mul(jn.d12, cosAngle * fringe));
		const V2 q5 = v2add(p1, v2mul(n12, hswAA));
		PosPair q; q.x0 = q0.x; q.y0 = q0.y; q.x1 = q1.x; q.y1 = q1.y;
		PosPair r; r.x0 = q2.x; r.y0 = q2.y; r.x1 = q3.x; r.y1 = q3.y;
		PosPair u; u.x0 = q4.x; u.y0 = q4.y; u.x1 = q5.x; u.y1 = q5.y;
		ColPair c; c.c0 = c0; c.c1 = color;
		ColPair d; d.c0 = color; d.c1 = c0;
		const uint32_t a = bi + 2u; // arcID
		Idx9 t; // the bevel as an arc of one segment (tri3 of the writer)
		if (L) {
			t.a = ((bi + 1u) & 0xFFFFu) | (a << 16); t.b = ((a + 2u) & 0xFFFFu) | (a << 16);
			t.c = ((a + 1u) & 0xFFFFu) | ((a + 3u) << 16); t.d = (a & 0xFFFFu) | ((a + 3u) << 16);
			t.e = (uint16_t)(a + 2u);
		} else {
			t.a = ((bi + 1u) & 0xFFFFu) | ((a + 2u) << 16); t.b = (a & 0xFFFFu) | (a << 16);
			t.c = ((a + 3u) & 0xFFFFu) | ((a + 1u) << 16); t.d = (a & 0xFFFFu) | ((a + 2u) << 16);
			t.e = (uint16_t)(a + 3u);
		}
		VGX_ST_GUARD(c0 ^ __float_as_uint(q.x0) ^ __float_as_uint(u.y1)) {
		*(PosPair*)pp = q;
		*(PosPair*)(pp + 16) = r;
		*(PosPair*)(pp + 32) = u;
		*(ColPair*)pc = c;
		*(ColPair*)(pc + 8) = d;
		*(ColPair*)(pc + 16) = d;
		TMPL_IDX_ON *(Idx9*)pio = t;
		}
		mine = L ? rails(bi, bi + 1, bi + 2, bi + 3) : rails(bi + 3, bi + 2, bi + 1, bi);
	}
	{
		// the bridge that ends at this join: from join j - 1's exit rails (elem_exit_rails), or -- join 0 -- the closing bridge from the last join's
		const uint32_t jm = j > 0 ? j - 1 : N - 1;
		const VgxJoin jp = vgx_join_dirs(dPrev2, dPrev, sideWidth);
		const uint32_t pb = R * jm + ibase;
		const Rails p = thin ? (jp.leftInner ? rails(pb, pb + 1, pb + 3, 0) : rails(pb + 3, pb + 1, pb, 0))
		                     : (jp.leftInner ? rails(pb, pb + 1, pb + 4, pb + 5) : rails(pb + 5, pb + 4, pb + 1, pb));
		const uint32_t kBridge = j > 0 ? kOwn - bridgeIdx : ownIdx + (N - 1u) * (bridgeIdx + ownIdx);
		char* pi = O.idx + (iOff + kBridge) * 2u;
		Idx6 t0; t0.a = (p.a & 0xFFFFu) | (p.b << 16); t0.b = (mine.b & 0xFFFFu) | (p.a << 16); t0.c = (mine.b & 0xFFFFu) | (mine.a << 16);
		Idx6 t1; t1.a = (p.b & 0xFFFFu) | (p.c << 16); t1.b = (mine.c & 0xFFFFu) | (p.b << 16); t1.c = (mine.c & 0xFFFFu) | (mine.b << 16);
		VGX_ST_GUARD(t0.a ^ t1.c) TMPL_IDX_ON {
		*(Idx6*)pi = t0;
		*(Idx6*)(pi + 12) = t1;
		if (!thin) {
			Idx6 t2; t2.a = (p.c & 0xFFFFu) | (p.d << 16); t2.b = (mine.d & 0xFFFFu) | (p.c << 16); t2.c = (mine.d & 0xFFFFu) | (mine.c << 16);
			*(Idx6*)(pi + 24) = t2;
		}
		}
	}
}

// Closed AA strokes with Round joins (stroker.cpp:1580-1691, closing bridge :1970-1984), the places from the per-step table: the element
// writes its own join -- inner pair, the arc's first pair, one pair per inner arc point (sincos), the arc's last pair; one fan triangle
// + fringe quad (9 indices) per arc segment -- and the bridge that ENDS at it (join 0: the closing bridge, at the end of the mesh's index
// range), whose far side is the previous join's exit rails: that join's place and inner side are its table word, its arc's point count the
// difference of the two places. da (the arc step, stroker.cpp:1398) is the mesh's, evaluated once per mesh and tile. Same values at the same
// places as elem_geometry + elem_emit (tests: VGX_TMPL_ROUND=0 byte for byte).
__device__ __forceinline__ bool tmpl_stroke_is_closed_round_aa(uint32_t kindWord)
{
	return VGX_MD_CLOSED(kindWord) != 0 && VGX_MD_KIND(kindWord) == VGX_MESH_STROKE_AA && VGX_MD_JOIN(kindWord) == VGX_JOIN_ROUND;
}
// Per-step table word pair of one element of a Round-join mesh (k_tmpl_round_sizes writes, the emit kernels read): its first vertex b and
// first index k inside the mesh, and what the bridge in front of it needs of the PREVIOUS element (join 0 of a closed stroke: of the last
// join): its vertex count nvPrev (-> its place and, 2 n + 4 or n + 2, its arc's point count) and its inner side.
//   x = b (16 bits) | prevInner << 16 | (nvPrev >> 1) << 17 (15 bits)        y = k (20 bits) | (nvPrev & 1) << 20
// Meshes hold <= 65 536 vertices (16-bit indices; an element's place is < 65 535, an element's size <= 65 534) and so < 2^19 indices; a mesh
// beyond that ends the call (the scan over the meshes), nothing reads its words. A closed AA stroke needs x only: its joins have 2 n + 4 (even)
// vertices, and k follows from b (k_j = 9 (b_j - 4 j) / 2 + 18 (j - 1): 9 indices per arc segment, one 18-index bridge per join in front).
__device__ __forceinline__ uint2 tmpl_round_word(uint32_t b, uint32_t k, uint32_t nvPrev, bool prevInner)
{
	return make_uint2((b & 0xFFFFu) | (prevInner ? 0x10000u : 0u) | (((nvPrev >> 1) & 0x7FFFu) << 17), (k & 0xFFFFFu) | ((nvPrev & 1u) << 20));
}
__device__ __forceinline__ void tmpl_round_unword(uint32_t x, uint32_t y, uint32_t* b, uint32_t* k, uint32_t* nvPrev, bool* prevInner)
{
	*b = x & 0xFFFFu; *k = y & 0xFFFFFu; *nvPrev = ((x >> 17) << 1) | ((y >> 20) & 1u); *prevInner = (x & 0x10000u) != 0;
}
struct TmplRoundPlace // what an element of a Round-join mesh takes from the per-step tables
{
	bool placed;       // the element belongs to a Round-join mesh
	uint32_t b, k;     // its first vertex / index inside the mesh
	uint32_t nvPrev;   // vertices of the element in front (join 0 of a closed stroke: of the LAST join): its place = b - nvPrev (join 0: nv - nvPrev)
	bool prevInner;    // that element's inner side (leftInner)
	uint32_t nv, ni;   // the mesh's vertices / indices (closing bridge)
	float da;          // the mesh's arc step
};
// OPEN: the template holds open strokes of this style (else every one is closed: no cap code, no cap sizes in the kernel)
template<bool OPEN>
__device__ __forceinline__ void tmpl_stroke_elem_round(const TmplOut& O, uint32_t kindWord, uint32_t N, uint32_t vOff, uint32_t iOff, uint32_t ibase, uint32_t color, float hsw, float hswAA, float fringe,
	uint32_t j, V2 p1, V2 dPrev, V2 d12, const TmplRoundPlace& rp)
{
	const bool closed = !OPEN || VGX_MD_CLOSED(kindWord) != 0;
	const uint32_t cap = VGX_MD_CAP(kindWord);
	// open strokes: the caps' sizes (elem_geometry) -- what lies in front of the joins
	const uint32_t H = (!closed && cap == VGX_CAP_ROUND) ? vgx_half_circle_points(rp.da) : 2u;
	const uint32_t capNv = cap == VGX_CAP_ROUND ? 2u * H : 4u, capNi = cap == VGX_CAP_ROUND ? 9u * H - 12u : 6u;
	// first index of the element's range (= of the bridge that ends at it), from its first vertex: 9 indices per arc segment of the joins in front
	// (2 n + 4 vertices each), one 18-index bridge per element in front but the first
	const uint32_t b = rp.b, bi = b + ibase; // index VALUES carry the assembly base, positions in the streams do not
	const uint32_t k = closed ? (j == 0 ? 0u : 9u * ((b - 4u * j) >> 1) + 18u * (j - 1u))
	                          : (j == 0 ? 0u : capNi + 9u * ((b - capNv - 4u * (j - 1u)) >> 1) + 18u * (j - 1u));
	// the previous element's exit rails (elem_exit_rails): a join's from its place, size and inner side; the first cap's are fixed
	Rails prev;
	{
		const uint32_t nvPrev = rp.nvPrev;
		const uint32_t pb = (j > 0 ? b : rp.nv) - nvPrev + ibase;
		prev = raa_join_exit(pb, (nvPrev - 4u) >> 1, rp.prevInner); // 2 n + 4 vertices (:1599)
		if (!closed && j == 1) { prev = raa_cap_first_exit(ibase, cap, H); }
	}
	char* pp = O.pos + (vOff + b) * 8u;
	char* pc = O.col + (vOff + b) * 4u;
	if (!closed && (j == 0 || j + 1 == N)) {
		raa_cap_emit(pp, pc, O.idx + (iOff + k) * 2u, cap, j == 0, bi, color, hsw, hswAA, fringe, p1, j == 0 ? d12 : dPrev, H, prev);
		return;
	}
	const VgxJoin jn = vgx_join_dirs(dPrev, d12, hswAA);
	const V2 n01 = jn.leftInner ? v2cw(jn.d01) : v2ccw(jn.d01);
	const V2 n12 = jn.leftInner ? v2cw(jn.d12) : v2ccw(jn.d12);
#ifdef VGX_EXP_FAKEARC /* measurement only: no atan2 pair */
	VgxArc arc; arc.a01 = n01.x; arc.arcDa = n12.y * 0.01f; arc.n = 2;
#else
	const VgxArc arc = vgx_round_join_arc(n01, n12, jn.leftInner, rp.da);
#endif
	// own join; the bridge that ENDS at it: in front of its own triangles, or -- join 0 of a closed stroke -- the closing bridge at the end of the mesh's indices
	char* pbridge = O.idx + (iOff + (j > 0 ? k : rp.ni - 18u)) * 2u;
	raa_join_emit(pp, pc, O.idx + (iOff + k + (j == 0 ? 0u : 18u)) * 2u, pbridge, bi, color, hsw, hswAA, p1, jn, arc, arc.n, prev);
}

// One element given its mesh's constants, its transformed vertex, its own edge direction and a way to get the mesh's other
// edge directions (dir(jj) = direction of the edge jj -> jj + 1, cyclic) and vertices (vtx(jj), general strokes only).
// PASS (GENERAL only): 0 = every element, 1 = everything but the general strokes, 2 = the general strokes only -- the tile loop runs
// pass 1 unrolled and pass 2 as a rolled loop, so that the general element body (~120 VGPRs of branches) is in the kernel once.
// KIND: what stroke styles the template holds: 0 = closed Miter AA / Thin only (the headline's kernel), 1 = + open Miter strokes with
// Butt / Square caps (tmpl_stroke_elem_open), 2 = + everything else (the general body; closed Bevel strokes take tmpl_stroke_elem_bevel there too),
// 3 = closed Miter and closed Bevel strokes only (tmpl_stroke_elem_bevel beside tmpl_stroke_elem: no general body in the kernel; with ROUND: + closed
// AA strokes with Round joins, tmpl_stroke_elem_round), 4 = 3 + OPEN AA strokes with Round joins (their caps).
#ifndef VGX_TMPL_K3_ROLLED
#define VGX_TMPL_K3_ROLLED 0 /* 1 (measured: bevel 2.22 -> 2.33 ms, round the same): one copy of the element routines, one element per trip */
#endif
#ifndef VGX_TMPL_RC_ROLLED
#define VGX_TMPL_RC_ROLLED 0 /* k_tmpl_emit_round_aa: 1 = the Round-join elements in a rolled second pass (what the 640-thread shape liked), 0 = inlined into the unrolled one */
#endif
#ifndef VGX_TMPL_BEVEL_FAST
#define VGX_TMPL_BEVEL_FAST 1 /* 0 (measurement): closed Bevel strokes through the general body */
#endif
template<int KIND, int PASS, class DF, class VF>
__device__ __forceinline__ void tmpl_elem_emit(const TmplOut& O, uint32_t j, uint32_t kindWord, uint32_t N, uint32_t vOff, uint32_t iOff, uint32_t ibase, uint32_t color, float f0, float f1,
	V2 p1, V2 d12, const DF& dir, const VF& vtx, float fringe, const vgx_draw* tdraw, const VgxTmplMesh* tmm, const TmplRoundPlace& rpl)
{
	const bool placed = rpl.placed;
	const uint32_t kind = VGX_MD_KIND(kindWord);
	const uint32_t jp1 = j > 0 ? j - 1 : N - 1;
	constexpr bool GENERAL = KIND == 2, OPEN = KIND == 1 || KIND == 2, BEVEL = KIND == 3 || KIND == 4;
	const bool openFast = OPEN && kind >= VGX_MESH_STROKE && tmpl_stroke_is_open_fast(kindWord);
	const bool general = GENERAL && kind >= VGX_MESH_STROKE && !openFast && !stroke_elem_is_simple(kind, VGX_MD_CLOSED(kindWord) != 0, VGX_MD_JOIN(kindWord));
	if (GENERAL && ((PASS == 1 && general) || (PASS == 2 && !general))) { return; }
	const bool closedBevel = (BEVEL || (general && VGX_TMPL_BEVEL_FAST)) && kind >= VGX_MESH_STROKE && tmpl_stroke_is_closed_bevel(kindWord);
	if (kind < VGX_MESH_STROKE) {
		V2 dPrev = d12;
		if (kind == VGX_MESH_FILL_AA) { dPrev = dir(jp1); }
		tmpl_fill_elem(O, kindWord, N, vOff, iOff, ibase, color, f0, j, p1, dPrev, d12);
	} else if (BEVEL && placed) { // (the kernel of closed strokes only: a Round-join mesh there is a closed AA one)
		tmpl_stroke_elem_round<KIND == 4>(O, kindWord, N, vOff, iOff, ibase, color, f0, f1, fringe, j, p1, (KIND == 4 && j == 0 && VGX_MD_CLOSED(kindWord) == 0) ? d12 : dir(jp1), d12, rpl);
	} else if (closedBevel) {
		const V2 dPrev = dir(jp1);
		const V2 dPrev2 = dir(jp1 > 0 ? jp1 - 1 : N - 1); // cyclic: element 0's previous join is the last one
		tmpl_stroke_elem_bevel(O, kindWord, N, vOff, iOff, ibase, color, f0, f1, fringe, j, p1, dPrev2, dPrev, d12);
	} else if (general) {
		const bool closed = VGX_MD_CLOSED(kindWord) != 0;
		const V2 dPrev = dir(jp1);
		V2 pPrev = p1, dPrev2 = dPrev, v0 = p1, v1 = p1;
		if (j > 0 && !(VGX_TMPL_ROUND_PREV_TABLE && placed)) { pPrev = vtx(jp1); dPrev2 = dir(jp1 > 0 ? jp1 - 1 : N - 1); } // the previous element's geometry: only when a bridge connects to it (Round-join meshes: from the per-step table)
		if (closed && j + 1 == N) { v0 = vtx(0u); v1 = vtx(N > 1 ? 1u : 0u); }      // join 0's inner side: only the closing bridge asks
		tmpl_stroke_general(O.pos, O.col, O.idx, kindWord, N, vOff, iOff, ibase, color, f0, f1, fringe, tdraw, j, p1, pPrev, d12, dPrev, dPrev2, v0, v1, placed, rpl.b, rpl.k, rpl.nvPrev, rpl.prevInner, placed ? rpl.da : -1.0f);
	} else if (openFast) {
		const V2 dPrev = dir(jp1);
		V2 dPrev2 = dPrev;
		if (j > 1) { dPrev2 = dir(jp1 - 1); }
		tmpl_stroke_elem_open(O, kindWord, N, vOff, iOff, ibase, color, f0, f1, tmm, j, p1, dPrev2, dPrev, d12);
	} else {
		const V2 dPrev = dir(jp1);
		const V2 dPrev2 = dir(jp1 > 0 ? jp1 - 1 : N - 1); // cyclic: element 0's previous join is the last one
		tmpl_stroke_elem(O, kindWord, N, vOff, iOff, ibase, color, f0, f1, j, p1, dPrev2, dPrev, d12);
	}
}

// VGX_TMPL_THREADS / VGX_TMPL_MAX_TILE (vgx_internal.h): threads per workgroup, elements per tile the LDS stages hold
// Draw-command assembly armed: the partition of the mesh sequence into vertex buffers / draw commands (vgx_assemble.hip) reads
// the WHOLE batch's mesh table and each mesh's draw; in template mode neither exists in memory, so this pass writes them
// (template record + instance offsets) before the assembly kernels run. k_tmpl_emit then adds each mesh's base to its indices.
__global__ __launch_bounds__(256) void k_tmpl_mtab(VgxTmplArgs A, vgx_mesh* mtab, VgxMeshDesc* mdesc)
{
	const uint64_t M = A.inst.num_meshes, total = A.total.num_meshes;
	if (blockIdx.x == 0 && threadIdx.x == 0 && !A.mplace) { A.totals->sizes = A.total; } // (Round joins: the scan over the meshes wrote them)
	for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t inst, vb, ib;
		uint32_t m;
		if (A.mplace) { // Round joins: places and sizes are this step's (per-step table, batch mesh order = k)
			inst = k / M;
			m = (uint32_t)(k - inst * M);
			if (A.iinfo) { // several classes: the instance that owns mesh k = the last one whose first mesh is <= k
				uint64_t lo = 0, hi = A.ninst;
				while (hi - lo > 1) {
					const uint64_t mid = (lo + hi) >> 1;
					if (A.iinfo[mid].m <= k) { lo = mid; } else { hi = mid; }
				}
				inst = lo;
				m = (uint32_t)(k - A.iinfo[lo].m) + A.iinfo[lo].cmesh0;
			}
			const VgxTmplMeshPlace q = A.mplace[k];
			vgx_mesh r = A.tmtab[m];
			r.first_vertex = q.v + (A.iplace ? A.iplace[2 * inst] : 0ull); r.first_index = q.i + (A.iplace ? A.iplace[2 * inst + 1] : 0ull); r.num_vertices = q.nv; r.num_indices = q.ni;
			r.draw += (uint32_t)(inst * A.period);
			mtab[k] = r;
			mdesc[k].draw = r.draw;
			continue;
		}
		if (A.iinfo) { // several classes: the instance that owns mesh k = the last one whose first mesh is <= k
			uint64_t lo = 0, hi = A.ninst;
			while (hi - lo > 1) {
				const uint64_t mid = (lo + hi) >> 1;
				if (A.iinfo[mid].m <= k) { lo = mid; } else { hi = mid; }
			}
			const VgxTmplInst ii = A.iinfo[lo];
			inst = lo; vb = ii.v; ib = ii.i;
			m = (uint32_t)(k - ii.m) + ii.cmesh0;
		} else {
			inst = k / M;
			m = (uint32_t)(k - inst * M);
			vb = inst * A.inst.num_vertices; ib = inst * A.inst.num_indices;
		}
		vgx_mesh r = A.tmtab[m];
		r.first_vertex += vb;
		r.first_index += ib;
		r.draw += (uint32_t)(inst * A.period);
		mtab[k] = r;
		mdesc[k].draw = r.draw;
	}
}
#ifndef VGX_TMPL_OCC
#define VGX_TMPL_OCC
#endif
// GENERAL: the template holds stroke meshes that are not closed Miter AA / Thin (open strokes, Bevel joins, non-AA): those take
// tmpl_stroke_general; the instantiation without them is the headline's kernel, unchanged.
// THREADS x MAXTILE: the workgroup shape (MAXTILE / THREADS elements per thread). The general instantiation (k_tmpl_emit_general) runs
// 4-wave workgroups over 2048-element tiles; k_tmpl_emit / k_tmpl_emit_open are VGX_TMPL_THREADS x VGX_TMPL_MAX_TILE.
#ifndef VGX_TMPL_G_THREADS
#define VGX_TMPL_G_THREADS 256
#endif
#define VGX_TMPL_G_TILE VGX_TMPL_GENERAL_TILE
// ROUND: the template holds Round-join meshes: the places of the instance, of its meshes and of every element of a Round-join mesh come from
// the per-step tables (k_tmpl_round_sizes / the scan over the meshes) instead of the template's closed forms.
template<int KIND, int THREADS, int MAXTILE, int ROUND = 0>
__device__ __forceinline__ void tmpl_emit_body(const VgxTmplArgs& A)
{
	constexpr bool GENERAL = KIND == 2;
	static_assert(ROUND == 0 || GENERAL || KIND == 3 || KIND == 4, "Round joins take the general element body, or -- closed AA strokes only -- tmpl_stroke_elem_round");
	constexpr int CH = MAXTILE / THREADS;
	__shared__ TmplDraw s_draw[VGX_TMPL_MAXM];
	__shared__ TmplRec s_rec[VGX_TMPL_MAXM];
	__shared__ float2 s_vtx[MAXTILE];
	__shared__ float2 s_dir[MAXTILE];
	__shared__ uint32_t s_status;
	__shared__ uint4 s_mi[ROUND ? VGX_TMPL_MAXM : 1];     // per mesh of the tile: first vertex / index inside the instance, vertices, indices (phase 0a -> 0b)
	__shared__ float4 s_rmesh[ROUND ? VGX_TMPL_MAXM : 1]; // per mesh of the tile: vertices, indices (bits), arc step, first table word of a Round-join mesh (bits; ~0: none)
	__shared__ float s_fringe[KIND >= 2 ? VGX_TMPL_MAXM : 1]; // per mesh of the tile: the draw's fringe (Bevel joins, Butt caps, thin strokes: kernels with those only)
	const uint32_t tid = threadIdx.x;
	// workgroup -> (instance, tile of the template), all workgroup-uniform (scalar loads)
	uint32_t inst32, t;
	TmplPlace P;
	uint64_t relemAt = 0; // ROUND: where the instance's table words begin (minus its class's first element number)
	if (A.wg) { // several classes: the tiles of an instance are its class's, the output places come from the per-instance table
		const uint2 w = A.wg[blockIdx.x];
		inst32 = w.x; t = w.y;
		const VgxTmplInst ii = A.iinfo[inst32];
		P.v = ii.v; P.i = ii.i; P.m = ii.m;
		if (ROUND) { P.v = A.iplace[2 * (uint64_t)inst32]; P.i = A.iplace[2 * (uint64_t)inst32 + 1]; relemAt = ii.rel; } // (several classes: the per-instance shape of the sizes pass)
	} else {
		inst32 = blockIdx.x / A.tiles_per_inst;
		t = blockIdx.x - inst32 * A.tiles_per_inst;
		P.v = (uint64_t)inst32 * A.inst.num_vertices; P.i = (uint64_t)inst32 * A.inst.num_indices; P.m = (uint64_t)inst32 * A.inst.num_meshes;
		if (ROUND) {
			if (A.iplace) { P.v = A.iplace[2 * (uint64_t)inst32]; P.i = A.iplace[2 * (uint64_t)inst32 + 1]; } // (mplace: places inside the instance)
			else { const VgxTmplMeshPlace* mp = A.mplace + (uint64_t)inst32 * A.inst.num_meshes; P.v = mp->v; P.i = mp->i; } // (mplace: places in the batch) the instance begins where its first mesh does
			relemAt = (uint64_t)inst32 * A.num_round_elems;
		}
	}
	const uint64_t inst = inst32;
	const VgxTmplTile tl = A.ttile[t];
	P.draw0 = (uint32_t)(inst * A.period); P.cmesh0 = tl.cmesh0; P.tdraws = A.tdraws + tl.cdraw0;
	const uint32_t x0 = t * A.tile;
	const uint32_t nel = tl.nel;
	const uint32_t mA = tl.mesh0 & 0x7FFFFFFFu;
	const bool firstWhole = (tl.mesh0 >> 31) != 0; // mesh mA begins in this tile (else in an earlier one)
	const uint32_t nm = tl.mesh_last - mA + 1;
	const uint32_t dA = tl.draw0, nd = tl.ndraws;
	const vgx_draw* idraws = A.draws + inst * A.period;
	const VgxTmplElem* telem = A.telem + x0;
	const uint32_t* meshBase = A.mesh_base ? A.mesh_base + P.m : nullptr; // the instance's meshes: indexed by (template mesh number - P.cmesh0)
	TmplOut O;
	O.pos = (char*)(A.pos + 2 * P.v);
	O.col = (char*)(A.color + P.v);
	O.idx = (char*)(A.idx + P.i);
	if (ROUND == 0 && blockIdx.x == 0 && tid == 0 && !A.mesh_base) { // (assembly armed: k_tmpl_mtab wrote them already; Round joins: the scan over the instances did) totals of the batch (the memset in front of this kernel zeroed them)
		A.totals->sizes = A.total;
	}
	// this instance's meshes in the per-step table: indexed by (template mesh number - P.cmesh0); P.cmesh0 = 0 (one class). Places inside the
	// instance are 32-bit (the stores use workgroup-uniform stream bases + 32-bit offsets): an instance beyond that ends the call
	const VgxTmplMeshPlace* mplace = ROUND ? A.mplace + P.m : nullptr; // (P.m = instance x meshes per instance for one class)
	auto minfoOf = [&](uint32_t m) {
		const VgxTmplMeshPlace q = mplace[m - P.cmesh0];
		const unsigned long long dv = A.iplace ? q.v : q.v - P.v, di = A.iplace ? q.i : q.i - P.i;
		if ((dv | di) >> 32) { set_status(A.totals, VGX_E_RANGE); }
		return make_uint4((uint32_t)dv, (uint32_t)di, q.nv, q.ni);
	};
	// table word of element number e among the template's Round-join elements (relemAt may be "negative" -- it wraps --: the sum is formed first, then the pointer)
	auto relemOf = [&](uint32_t e) { return A.relem + (relemAt + (uint64_t)e); };

	if (nm > VGX_TMPL_MAXM || nd > VGX_TMPL_MAXM) {
		// workgroup-uniform. Many tiny meshes (or many draws without a mesh) in one tile: the draw records are verified in a loop,
		// every lane fetches its own records and neighbours
		for (uint32_t k = tid; k < nd; k += THREADS) { (void)tmpl_load_draw(A, idraws, P.tdraws, dA + k); }
		for (uint32_t s = tid; s < nel; s += THREADS) {
			const VgxTmplElem er = telem[s];
			const VgxTmplMesh tm = A.tmesh[er.mesh];
			const TmplDraw dr = tmpl_load_draw(A, idraws, P.tdraws, tm.drawk);
			const TmplXf xf = tmpl_draw_xf(&dr);
			const float2* vt = A.tpoly + tm.poly_first;
			const uint32_t kind = VGX_MD_KIND(tm.kind);
			const uint32_t j = er.jq & 0xFFFFu, N = tm.n;
			float f0 = tm.f0;
			if (kind == VGX_MESH_FILL_AA) { f0 = tmpl_fill_aa(xf, vt[0], vt[1], vt[2], tm.f0); }
			auto dir = [&](uint32_t jj) { return v2dir(tmpl_xf(xf, vt[jj]), tmpl_xf(xf, vt[jj + 1 < N ? jj + 1 : 0u])); };
			auto vtx = [&](uint32_t jj) { return tmpl_xf(xf, vt[jj]); };
			uint32_t vOff = tm.v_off, iOff = tm.i_off;
			TmplRoundPlace rpl;
			rpl.placed = false; rpl.b = 0; rpl.k = 0; rpl.nvPrev = 0; rpl.prevInner = false; rpl.nv = 0; rpl.ni = 0; rpl.da = 0.0f;
			if (ROUND) {
				const uint4 mi = minfoOf(er.mesh);
				vOff = mi.x; iOff = mi.y;
				if (j == 0 && A.meshes_out) { tmpl_mesh_out_placed(A, P, er.mesh, mi); }
				if (tm.pad[1] != 0) {
					const uint2 bk = *relemOf(__float_as_uint(tm.l2[1]) + j);
					rpl.placed = true;
					tmpl_round_unword(bk.x, bk.y, &rpl.b, &rpl.k, &rpl.nvPrev, &rpl.prevInner);
					rpl.nv = mi.z; rpl.ni = mi.w;
					rpl.da = tm.l2[0];
				}
			} else if (j == 0 && A.meshes_out) { tmpl_mesh_out(A, P, er.mesh); }
			const uint32_t ibase = meshBase ? meshBase[er.mesh - P.cmesh0] : 0u;
			tmpl_elem_emit<KIND, 0>(O, j, tm.kind, N, vOff, iOff, ibase, kind < VGX_MESH_STROKE ? dr.fill_color : dr.stroke_color, f0, tm.f1, tmpl_xf(xf, vt[j]), dir(j), dir, vtx,
				__uint_as_float(tm.pad[0]), P.tdraws + tm.drawk, A.tmesh + er.mesh, rpl);
		}
		return;
	}
#ifdef VGX_TMPL_PROFILE
	const unsigned long long tp0 = wall_clock64();
#define TMPL_PROF(i) do { if (tid == 0) { atomicAdd(&A.totals->prof[i], wall_clock64() - tp0); } } while (0)
#else
#define TMPL_PROF(i)
#endif
	// ---- phase 0a: every load the workgroup needs, requested at once
	if (ROUND && tid < nm) { s_mi[tid] = minfoOf(mA + tid); } // requested FIRST and parked in LDS as soon as it is there (the loads below stay in flight): its registers are free again before the element and mesh records arrive
	VgxTmplElem er[CH];
	uint32_t rb[ROUND ? CH : 1], rk[ROUND ? CH : 1]; // ROUND: elements of Round-join meshes: their table words (else rb = ~0)
#pragma unroll
	for (int c = 0; c < CH; ++c) {
		const uint32_t s = (uint32_t)c * THREADS + tid; // interleaved: the tile's stroke chunks (the heavier ones, at the tile's end) spread over the waves
		er[c].mesh = mA; er[c].jq = 0; er[c].lx = 0.0f; er[c].ly = 0.0f;
		if (s < nel) { er[c] = telem[s]; }
	}
	VgxTmplMesh tm;
	memset(&tm, 0, sizeof(tm));
	tm.n = 3; tm.drawk = dA; tm.kind = VGX_MESH_FILL;
	uint32_t ibase = 0;
	uint4 mi = make_uint4(0u, 0u, 0u, 0u);
	if (tid < nm) {
		tm = A.tmesh[mA + tid];
		if (meshBase) { ibase = meshBase[mA - P.cmesh0 + tid]; }
	}
	if (tid < nd) { s_draw[tid] = tmpl_load_draw(A, idraws, P.tdraws, dA + tid); }
	if (tid == 0) { s_status = A.totals->status; } // an earlier workgroup may have found the batch stale; ONE value for the whole workgroup (the exit below must be uniform)
	__syncthreads();
	const uint32_t status = s_status;
	// ---- phase 0b: per-mesh records
	if (tid < nm) {
		if (ROUND) { mi = s_mi[tid]; tm.v_off = mi.x; tm.i_off = mi.y; } // this instance's places
		const uint32_t kind = VGX_MD_KIND(tm.kind);
		const TmplDraw* d = &s_draw[tm.drawk - dA];
		TmplRec r;
		r.ibase = ibase; r.n = tm.n; r.v_off = tm.v_off; r.i_off = tm.i_off;
		r.kind = (tm.kind & 0xFFFFu) | ((tm.drawk - dA) << 16); r.color = kind < VGX_MESH_STROKE ? d->fill_color : d->stroke_color; r.f0 = tm.f0; r.f1 = tm.f1;
		if (KIND >= 2) { s_fringe[tid] = __uint_as_float(tm.pad[0]); }
		if (ROUND) { // Round-join meshes: the mesh's sizes (closing bridge), arc step (stroker.cpp:1398; the closed-stroke kernel's routine) and first table word, once per mesh
			const bool rj = tm.pad[1] != 0; // (both from the mesh record: no load behind the first barrier)
			s_rmesh[tid] = make_float4(__uint_as_float(mi.z), __uint_as_float(mi.w), rj ? tm.l2[0] : 0.0f, rj ? tm.l2[1] : __uint_as_float(~0u));
		}
		if (kind == VGX_MESH_FILL_AA) { r.f0 = tmpl_fill_aa(tmpl_draw_xf(d), make_float2(tm.l0[0], tm.l0[1]), make_float2(tm.l1[0], tm.l1[1]), make_float2(tm.l2[0], tm.l2[1]), tm.f0); }
		s_rec[tid] = r;
		// the caller's mesh table for the meshes that BEGIN in this tile (every mesh of the range but possibly the first)
		if ((tid > 0 || firstWhole) && A.meshes_out && status == VGX_OK) {
			if (ROUND) { tmpl_mesh_out_placed(A, P, mA + tid, mi); } else { tmpl_mesh_out(A, P, mA + tid); }
		}
	}
	__syncthreads();
	TMPL_PROF(0);
	if (status != VGX_OK) { // workgroup-uniform
		return;
	}
	if (ROUND) {
		// elements of Round-join meshes: their table words (place; the size and inner side of the element in front) -- requested here, used in phase 3
#pragma unroll
		for (int c = 0; c < CH; ++c) {
			const uint32_t s = (uint32_t)c * THREADS + tid;
			rb[c] = ~0u; rk[c] = 0;
			if (s < nel) {
				const uint32_t e0 = __float_as_uint(s_rmesh[er[c].mesh - mA].w);
				if (e0 != ~0u) { // (x is never ~0: a place is < 65 535)
					const uint2* w = relemOf(e0 + (er[c].jq & 0xFFFFu));
					if (GENERAL) { const uint2 bk = *w; rb[c] = bk.x; rk[c] = bk.y; } else { rb[c] = w->x; }
				}
			}
		}
	}
	// ---- phase 1: own vertex, transformed once
	V2 p1[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) {
		const uint32_t s = (uint32_t)c * THREADS + tid;
		p1[c] = v2(0.0f, 0.0f);
		if (s < nel) {
			const TmplRec* r = &s_rec[er[c].mesh - mA];
			p1[c] = tmpl_xf(tmpl_draw_xf(&s_draw[TMPL_REC_DK(r)]), make_float2(er[c].lx, er[c].ly));
			s_vtx[er[c].jq >> 16] = make_float2(p1[c].x, p1[c].y);
		}
	}
	__syncthreads();
	TMPL_PROF(1);
	// transformed vertex jj of mesh `mesh` whose vertex 0 sits at tile position q0 (negative: in front of the tile): LDS, or -- the
	// vertex belongs to another tile -- L2 + transform
	auto vtxAt = [&](uint32_t mesh, const TmplRec* rp, int q0, uint32_t jj) {
		const uint32_t qq = (uint32_t)(q0 + (int)jj);
		if (qq < nel) { const float2 v = s_vtx[qq]; return v2(v.x, v.y); }
		return tmpl_xf(tmpl_draw_xf(&s_draw[TMPL_REC_DK(rp)]), A.tpoly[A.tmesh[mesh].poly_first + jj]);
	};
	// ---- phase 2: own edge direction vec2Dir(p[j], p[j + 1]) (stroker.cpp:31-38), once per element
	V2 d12[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) {
		const uint32_t s = (uint32_t)c * THREADS + tid;
		d12[c] = v2(0.0f, 0.0f);
		if (s < nel) {
			const TmplRec* rp = &s_rec[er[c].mesh - mA];
			const uint32_t j = er[c].jq & 0xFFFFu, N = rp->n;
			const int q0 = (int)(er[c].jq >> 16) - (int)j;
			d12[c] = v2dir(p1[c], vtxAt(er[c].mesh, rp, q0, j + 1 < N ? j + 1 : 0u));
			s_dir[er[c].jq >> 16] = make_float2(d12[c].x, d12[c].y);
		}
	}
	__syncthreads();
	TMPL_PROF(2);
	// ---- phase 3: the element
	auto element = [&](auto passTag, uint32_t s, const VgxTmplElem& e, V2 pv, V2 dv, bool placed, uint32_t wx, uint32_t wy) {
		TmplRoundPlace rpl; // (wx, wy: the element's table words, tmpl_round_word)
		rpl.placed = placed;
		tmpl_round_unword(wx, wy, &rpl.b, &rpl.k, &rpl.nvPrev, &rpl.prevInner);
		rpl.nv = 0; rpl.ni = 0; rpl.da = 0.0f;
		if (s < nel) {
			const uint32_t mesh = e.mesh;
			const TmplRec* rp = &s_rec[mesh - mA];
			const uint32_t j = e.jq & 0xFFFFu, N = rp->n;
			const int q0 = (int)(e.jq >> 16) - (int)j;
			auto dir = [&](uint32_t jj) {
				const uint32_t qq = (uint32_t)(q0 + (int)jj);
				if (qq < nel) { const float2 v = s_dir[qq]; return v2(v.x, v.y); }
				return v2dir(vtxAt(mesh, rp, q0, jj), vtxAt(mesh, rp, q0, jj + 1 < N ? jj + 1 : 0u)); // the edge belongs to another tile
			};
			auto vtx = [&](uint32_t jj) { return vtxAt(mesh, rp, q0, jj); };
			const float fringe = KIND >= 2 ? s_fringe[mesh - mA] : 0.0f;
			if (ROUND && placed) { const float4 rm = s_rmesh[mesh - mA]; rpl.nv = __float_as_uint(rm.x); rpl.ni = __float_as_uint(rm.y); rpl.da = rm.z; }
			const vgx_draw* tdraw = P.tdraws;
			if (GENERAL && decltype(passTag)::value == 2) { tdraw = P.tdraws + (dA + TMPL_REC_DK(rp)); }
			tmpl_elem_emit<KIND, decltype(passTag)::value>(O, j, rp->kind & 0xFFFFu, N, rp->v_off, rp->i_off, rp->ibase, rp->color, rp->f0, rp->f1, pv, dv, dir, vtx, fringe, tdraw, A.tmesh + mesh,
				rpl);
		}
	};
	if ((KIND == 3 || KIND == 4) && VGX_TMPL_K3_ROLLED) {
		// the kernels of closed strokes only: ONE copy of the element routines, one element per trip -- 4 x (fill + Miter + Bevel + Round)
		// inlined is 45-67 KB of code per kernel, against an instruction cache of 64 KB shared by two CUs
#pragma unroll 1
		for (int c = 0; c < CH; ++c) {
			VgxTmplElem e = er[0]; V2 pv = p1[0], dv = d12[0];
			uint32_t pb = ROUND ? rb[0] : ~0u, pk = ROUND ? rk[0] : 0u;
#pragma unroll
			for (int k = 1; k < CH; ++k) { if (c == k) { e = er[k]; pv = p1[k]; dv = d12[k]; if (ROUND) { pb = rb[k]; pk = rk[k]; } } }
			element(std::integral_constant<int, 0>(), (uint32_t)c * THREADS + tid, e, pv, dv, ROUND != 0 && pb != ~0u, pb, pk);
		}
	} else {
#pragma unroll
	for (int c = 0; c < CH; ++c) {
		if (ROUND && !GENERAL) {
			const uint32_t s = (uint32_t)c * THREADS + tid;
			if (!VGX_TMPL_RC_ROLLED) { element(std::integral_constant<int, 0>(), s, er[c], p1[c], d12[c], rb[c] != ~0u, rb[c], rk[c]); }
			else if (rb[c] == ~0u) { element(std::integral_constant<int, 0>(), s, er[c], p1[c], d12[c], false, 0u, 0u); }
		}
		else { element(std::integral_constant<int, GENERAL ? 1 : 0>(), (uint32_t)c * THREADS + tid, er[c], p1[c], d12[c], false, 0u, 0u); }
	}
	if (ROUND && !GENERAL && VGX_TMPL_RC_ROLLED) { // the Round-join elements of the tile, one per trip: tmpl_stroke_elem_round exists once
#pragma unroll 1
		for (int c = 0; c < CH; ++c) {
			VgxTmplElem e = er[0]; V2 pv = p1[0], dv = d12[0];
			uint32_t pb = rb[0], pk = rk[0];
#pragma unroll
			for (int k = 1; k < CH; ++k) { if (c == k) { e = er[k]; pv = p1[k]; dv = d12[k]; pb = rb[k]; pk = rk[k]; } }
			if (pb != ~0u) { element(std::integral_constant<int, 0>(), (uint32_t)c * THREADS + tid, e, pv, dv, true, pb, pk); }
		}
	}
	}
	if (GENERAL) { // the general strokes of the tile, one element per trip: the body exists once
#pragma unroll 1
		for (int c = 0; c < CH; ++c) {
			VgxTmplElem e = er[0]; V2 pv = p1[0], dv = d12[0];
			uint32_t pb = ROUND ? rb[0] : 0u, pk = ROUND ? rk[0] : 0u;
#pragma unroll
			for (int k = 1; k < CH; ++k) { if (c == k) { e = er[k]; pv = p1[k]; dv = d12[k]; if (ROUND) { pb = rb[k]; pk = rk[k]; } } }
			element(std::integral_constant<int, 2>(), (uint32_t)c * THREADS + tid, e, pv, dv, ROUND != 0 && pb != ~0u, pb, pk);
		}
	}
	TMPL_PROF(3);
#ifdef VGX_TMPL_PROFILE
	__builtin_amdgcn_s_waitcnt(0); // vmcnt(0): the wave's stores are out
	TMPL_PROF(4);
	if (tid == 0) { atomicAdd(&A.totals->prof[5], 1ull); }
#endif
}

#ifdef VGX_TMPL_MINWAVES
__global__ __launch_bounds__(VGX_TMPL_THREADS, VGX_TMPL_MINWAVES) void k_tmpl_emit(VgxTmplArgs A)
#else
__global__ __launch_bounds__(VGX_TMPL_THREADS) VGX_TMPL_OCC void k_tmpl_emit(VgxTmplArgs A)
#endif
{
	tmpl_emit_body<0, VGX_TMPL_THREADS, VGX_TMPL_MAX_TILE>(A);
}
// the same shape with the open-stroke routine: templates whose only non-closed strokes are open Miter ones with Butt / Square caps
__global__ __launch_bounds__(VGX_TMPL_THREADS) void k_tmpl_emit_open(VgxTmplArgs A)
{
	tmpl_emit_body<1, VGX_TMPL_THREADS, VGX_TMPL_MAX_TILE>(A);
}
// the instantiation with the general stroke body: 256 threads x 2048-element tiles (eight elements per thread, ~145 VGPRs, three
// waves per SIMD): same-box A/B on the Tiger with Bevel joins 3.53 ms against 4.20 (256 x 1024), 3.80 (512 x 2048), 4.85 (512 x 1024)
#ifndef VGX_TMPL_G_MINWAVES
#define VGX_TMPL_G_MINWAVES 3
#endif
// closed strokes with Miter and Bevel joins only: the headline kernel's shape with the Bevel routine beside the Miter one
__global__ __launch_bounds__(VGX_TMPL_THREADS) void k_tmpl_emit_bevel(VgxTmplArgs A)
{
	tmpl_emit_body<3, VGX_TMPL_THREADS, VGX_TMPL_MAX_TILE>(A);
}
__global__ __launch_bounds__(VGX_TMPL_G_THREADS, VGX_TMPL_G_MINWAVES) void k_tmpl_emit_general(VgxTmplArgs A)
{
	tmpl_emit_body<2, VGX_TMPL_G_THREADS, VGX_TMPL_G_TILE>(A);
}

// ---- Round joins: the instantiation with the per-step places, and the kernels that make them --------------------------------
#ifndef VGX_TMPL_R_MINWAVES
#define VGX_TMPL_R_MINWAVES 3
#endif
__global__ __launch_bounds__(VGX_TMPL_G_THREADS, VGX_TMPL_R_MINWAVES) void k_tmpl_emit_round(VgxTmplArgs A)
{
	tmpl_emit_body<2, VGX_TMPL_G_THREADS, VGX_TMPL_G_TILE, 1>(A);
}
// templates whose strokes are all closed (Miter / Bevel joins, AA Round joins): the headline kernel's shape with the three closed routines
// What this kernel waits for is the third workgroup per CU the Bevel kernel (78 VGPRs) has: at 85-90 VGPRs (five waves per SIMD) only two
// 8-wave workgroups fit. Same box: 512 x 2048 at 87 VGPRs 3.47 ms; 640 x 2560 (two 10-wave workgroups = all five waves) with the
// Round-join elements in a rolled second pass 3.35; forced to 80 VGPRs with 20-84 bytes of scratch 3.95-4.14; and, once the per-step
// mesh places are parked in LDS in front of the other phase-0 loads (their registers free again before the records arrive), 80 VGPRs
// WITHOUT scratch, 512 x 2048, six waves: 3.03 against 3.22 -- the shipped shape
__global__ __launch_bounds__(VGX_TMPL_RC_THREADS, VGX_TMPL_RC_WAVES) void k_tmpl_emit_round_aa(VgxTmplArgs A)
{
	tmpl_emit_body<3, VGX_TMPL_RC_THREADS, VGX_TMPL_RC_TILE, 1>(A);
}
// the same with OPEN strokes of that style among them (Butt / Square / Round caps): the cap code costs the registers of the third workgroup
// per CU (80 VGPRs would spill): five waves per SIMD
__global__ __launch_bounds__(VGX_TMPL_RC_THREADS, 5) void k_tmpl_emit_round_aa_open(VgxTmplArgs A)
{
	tmpl_emit_body<4, VGX_TMPL_RC_THREADS, VGX_TMPL_RC_TILE, 1>(A);
}

// Sizes: one wave per (instance, Round-join mesh). Lane = element: its vertex and both neighbours through transformPos2D with the
// instance's matrix, the two edge directions, elem_geometry -- the very functions on the very inputs k_tmpl_emit_round's phases 1 - 3
// evaluate (so: the same bits, the arcs counted = the arcs emitted) --, a running prefix over the mesh's elements -> every element's
// first vertex / index inside its mesh (relem), the mesh's totals (rsz).
// what a wave needs of one Round-join mesh of one instance
struct TmplRoundRec // 48 bytes
{
	TmplXf xf;                 // the instance's transform of the mesh's draw
	uint32_t poly_first, n, kind, elem0;
	float hsw, hswAA, da; uint32_t pad;
};
// one mesh: lanes = elements, 64 per trip
__device__ __forceinline__ void tmpl_round_sizes_mesh(const VgxTmplArgs& A, uint64_t inst, uint32_t r, const TmplRoundRec& rc, uint32_t lane, unsigned long long* meshV, unsigned long long* meshI,
	uint64_t relemAt, bool writeRsz = true) // relemAt: where the instance's table words begin (one class: instance x elements per instance)
{
	const TmplXf xf = rc.xf;
	const float2* vt = A.tpoly + rc.poly_first;
	const uint32_t N = rc.n;
	MeshCtxT<TmplVtx01> mc;
	mc.kind = VGX_MD_KIND(rc.kind); mc.closed = VGX_MD_CLOSED(rc.kind) != 0; mc.cap = VGX_MD_CAP(rc.kind); mc.join = VGX_MD_JOIN(rc.kind);
	mc.N = N; mc.hsw = rc.hsw; mc.hswAA = rc.hswAA; mc.fringe = 0.0f; // (the fringe: thin strokes only, never a Round-join mesh)
	mc.dr = A.tdraws; mc.da = rc.da; // (the arc step: in the mesh record since the template was built; dr is not read)
	mc.vtx.x0 = 0.0f; mc.vtx.y0 = 0.0f; mc.vtx.x1 = 0.0f; mc.vtx.y1 = 0.0f;
	uint2* out = A.relem + (relemAt + (uint64_t)rc.elem0); // (relemAt wraps for all but a template's first class: the sum first)
	unsigned long long runV = 0, runI = 0;
	uint32_t carryNv = 0; bool carryInner = false; // the element in front of the chunk (wave-uniform)
	for (uint32_t j0 = 0; j0 < N; j0 += 64) {
		const uint32_t j = j0 + lane;
		uint32_t nv = 0, ni = 0;
		bool inner = false;
		if (j < N) {
			const V2 p1 = tmpl_xf(xf, vt[j]);
			const V2 pn = tmpl_xf(xf, vt[j + 1 < N ? j + 1 : 0u]);
			const V2 pp = tmpl_xf(xf, vt[j > 0 ? j - 1 : N - 1]);
			mc.j = j;
			const Elem e = elem_geometry(mc, p1, v2dir(pp, p1), v2dir(p1, pn));
			nv = e.nv; ni = elem_total_indices(mc, e); inner = e.leftInner;
		}
		unsigned long long v = nv, i = ni;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const unsigned long long tv = __shfl_up(v, d), ti = __shfl_up(i, d);
			if (lane >= (uint32_t)d) { v += tv; i += ti; }
		}
		// the element in front: the lane below, or the last lane of the chunk before (join 0: patched below)
		uint32_t nvP = __shfl_up(nv, 1);
		int inP = __shfl_up((int)inner, 1);
		if (lane == 0) { nvP = carryNv; inP = (int)carryInner; }
		if (j < N) { out[j] = tmpl_round_word((uint32_t)(runV + v - nv), (uint32_t)(runI + i - ni), nvP, inP != 0); }
		runV += __shfl(v, 63); runI += __shfl(i, 63);
		const uint32_t lastLane = (N - j0 < 64u ? N - j0 : 64u) - 1u;
		carryNv = __shfl(nv, (int)lastLane); carryInner = __shfl((int)inner, (int)lastLane) != 0;
	}
	if (lane == 0) {
		if (writeRsz) { // (the per-instance shape keeps the sizes in LDS)
			const uint64_t g = inst * A.num_round + r;
			A.rsz[2 * g] = runV; A.rsz[2 * g + 1] = runI;
		}
		if (mc.closed) { out[0] = tmpl_round_word(0u, 0u, carryNv, carryInner); } // join 0: the closing bridge starts at the LAST join (which is now known)
	}
	*meshV = runV; *meshI = runI; // (wave-uniform)
}
__device__ __forceinline__ TmplRoundRec tmpl_round_rec(const VgxTmplArgs& A, uint64_t inst, uint32_t r, const vgx_draw* tdraws) // r: number in trmesh; tdraws: the class representative's records
{
	const VgxTmplRoundMesh rm = A.trmesh[r];
	const VgxTmplMesh tm = A.tmesh[rm.mesh];
	const TmplDraw dr = tmpl_load_draw(A, A.draws + inst * A.period, tdraws, tm.drawk); // (verified: a stale or non-finite record ends the call like in the emit kernel)
	TmplRoundRec rc;
	rc.xf = tmpl_draw_xf(&dr);
	rc.poly_first = tm.poly_first; rc.n = tm.n; rc.kind = tm.kind; rc.elem0 = rm.elem0;
	rc.hsw = tm.f0; rc.hswAA = tm.f1; rc.da = tm.l2[0]; rc.pad = 0;
	return rc;
}
// One workgroup per INSTANCE (templates of up to VGX_TMPL_ROUND_MAXR Round-join meshes -- the Tiger has 110): the meshes' records and the
// instance's transforms are fetched by all threads at once and parked in LDS (one memory round trip for the instance instead of three
// dependent ones per mesh), then every wave takes every fourth mesh. One wave per mesh (the kernel below) spends its life in those round
// trips: Tiger x 10k, same box, the whole sizes stage (this + the scan over the meshes) 0.50 -> 0.41 ms.
#ifndef VGX_TMPL_ROUND_MAXR
#define VGX_TMPL_ROUND_MAXR 640
#endif
__global__ __launch_bounds__(256) void k_tmpl_round_sizes_inst(VgxTmplArgs A)
{
	extern __shared__ TmplRoundRec s_rc[]; // [num_round] (dynamic: a template of a hundred such meshes takes 5 KB, not the 30 KB of the limit)
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const uint64_t inst = blockIdx.x;
	// the instance's class (one class: the whole template): its Round-join meshes trmesh[R0, R0 + R), its meshes [cmesh0, cmesh0 + M) of the
	// template, its saved draw records, and where the instance's rows of the per-step tables begin
	uint32_t R = A.num_round, R0 = 0, M = (uint32_t)A.inst.num_meshes, cmesh0 = 0;
	const vgx_draw* tdraws = A.tdraws;
	uint64_t relemAt = inst * A.num_round_elems, mplaceAt = inst * M;
	if (A.iinfo) {
		const VgxTmplInst ii = A.iinfo[inst];
		const VgxTmplClass c0 = A.cls[ii.cls], c1 = A.cls[ii.cls + 1];
		R0 = c0.pad[1]; R = c1.pad[1] - R0; cmesh0 = c0.mesh0; M = c1.mesh0 - cmesh0;
		tdraws = A.tdraws + (uint64_t)ii.cls * A.period;
		relemAt = ii.rel; mplaceAt = ii.m;
	}
	for (uint32_t r = threadIdx.x; r < R; r += 256) { s_rc[r] = tmpl_round_rec(A, inst, R0 + r, tdraws); }
	__syncthreads();
	for (uint32_t r = wave; r < R; r += 4) {
		unsigned long long mv, mi;
		tmpl_round_sizes_mesh(A, inst, r, s_rc[r], lane, &mv, &mi, relemAt, false);
		if (lane == 0) { // the mesh's sizes stay here for the places below (the record's transform is not needed any more)
			s_rc[r].xf.m0 = __uint_as_float(vgx_sat_nv(mv)); s_rc[r].xf.m1 = __uint_as_float(vgx_sat_ni(mi)); s_rc[r].xf.m2 = __uint_as_float(mv > 65536ull ? 1u : 0u);
		}
	}
	__syncthreads();
	// every mesh's place INSIDE the instance (the template's sizes for the meshes without Round joins, the counted ones for the others) and the
	// instance's totals: the scan that follows runs over the instances, not over every mesh of the batch
	__shared__ unsigned long long s_wv[4], s_wi[4], s_runV, s_runI;
	__shared__ uint32_t s_big;
	if (threadIdx.x == 0) { s_runV = 0; s_runI = 0; s_big = 0; }
	__syncthreads();
	VgxTmplMeshPlace* mp = A.mplace + mplaceAt;
	for (uint32_t m0 = 0; m0 < M; m0 += 256) {
		const uint32_t m = m0 + threadIdx.x;
		uint32_t nv = 0, ni = 0;
		if (m < M) {
			const uint2 ts = A.tmsz[cmesh0 + m];
			if (ts.x >> 31) {
				const TmplRoundRec* rc = &s_rc[(ts.x & 0x7FFFFFFFu) - R0];
				nv = __float_as_uint(rc->xf.m0); ni = __float_as_uint(rc->xf.m1);
				if (__float_as_uint(rc->xf.m2)) { s_big = 1u; } // what OpMeshOffsets reports for such a mesh (16-bit indices)
			} else { nv = ts.x; ni = ts.y; }
		}
		unsigned long long v = nv, i = ni;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const unsigned long long tv = __shfl_up(v, d), ti = __shfl_up(i, d);
			if (lane >= (uint32_t)d) { v += tv; i += ti; }
		}
		if (lane == 63u) { s_wv[wave] = v; s_wi[wave] = i; }
		__syncthreads();
		unsigned long long baseV = s_runV, baseI = s_runI, totV = 0, totI = 0;
#pragma unroll
		for (uint32_t w = 0; w < 4; ++w) { if (w < wave) { baseV += s_wv[w]; baseI += s_wi[w]; } totV += s_wv[w]; totI += s_wi[w]; }
		if (m < M) { VgxTmplMeshPlace q; q.v = baseV + v - nv; q.i = baseI + i - ni; q.nv = nv; q.ni = ni; q.pad[0] = 0; q.pad[1] = 0; mp[m] = q; }
		__syncthreads();
		if (threadIdx.x == 0) { s_runV += totV; s_runI += totI; }
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		A.itot[2 * inst] = s_runV; A.itot[2 * inst + 1] = s_runI;
		if (s_big) { set_status(A.totals, VGX_E_MESH_TOO_LARGE); }
	}
}
// one wave per (instance, mesh), four to a workgroup: templates with more Round-join meshes than the LDS table holds
__global__ __launch_bounds__(256) void k_tmpl_round_sizes(VgxTmplArgs A)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t R = A.num_round;
	const uint64_t pairs = A.ninst * (uint64_t)R;
	const uint64_t g = (uint64_t)blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	if (g >= pairs) { return; }
	const uint64_t inst = g / R;
	const uint32_t r = (uint32_t)(g - inst * R);
	unsigned long long mv, mi;
	tmpl_round_sizes_mesh(A, inst, r, tmpl_round_rec(A, inst, r, A.tdraws), lane, &mv, &mi, inst * A.num_round_elems);
}

// The same for templates of LONG Round-join meshes (a polyline of a thousand segments: one wave per mesh leaves the GPU to a few thousand
// waves of sixteen sequential chunks each): one 256-thread workgroup per (instance, mesh), 256 elements per trip, the waves' sums through LDS.
__global__ __launch_bounds__(256) void k_tmpl_round_sizes_block(VgxTmplArgs A)
{
	__shared__ unsigned long long s_wv[4], s_wi[4], s_runV, s_runI;
	__shared__ uint32_t s_wlastNv[4], s_wlastIn[4], s_lastNv, s_lastIn;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t R = A.num_round;
	const uint64_t g = blockIdx.x;
	const uint64_t inst = g / R;
	const uint32_t r = (uint32_t)(g - inst * R);
	const VgxTmplRoundMesh rm = A.trmesh[r];
	const VgxTmplMesh tm = A.tmesh[rm.mesh];
	const TmplDraw dr = tmpl_load_draw(A, A.draws + inst * A.period, A.tdraws, tm.drawk);
	const TmplXf xf = tmpl_draw_xf(&dr);
	const float2* vt = A.tpoly + tm.poly_first;
	const uint32_t N = tm.n;
	MeshCtxT<TmplVtx01> mc;
	mc.kind = VGX_MD_KIND(tm.kind); mc.closed = VGX_MD_CLOSED(tm.kind) != 0; mc.cap = VGX_MD_CAP(tm.kind); mc.join = VGX_MD_JOIN(tm.kind);
	mc.N = N; mc.hsw = tm.f0; mc.hswAA = tm.f1; mc.fringe = 0.0f;
	mc.dr = A.tdraws + tm.drawk; mc.da = tm.l2[0]; // (the arc step: in the mesh record since the template was built)
	mc.vtx.x0 = 0.0f; mc.vtx.y0 = 0.0f; mc.vtx.x1 = 0.0f; mc.vtx.y1 = 0.0f;
	uint2* out = A.relem + inst * A.num_round_elems + rm.elem0;
	if (tid == 0) { s_runV = 0; s_runI = 0; s_lastNv = 0; s_lastIn = 0; }
	__syncthreads();
	for (uint32_t j0 = 0; j0 < N; j0 += 256) {
		const uint32_t j = j0 + tid;
		uint32_t nv = 0, ni = 0;
		bool inner = false;
		if (j < N) {
			const V2 p1 = tmpl_xf(xf, vt[j]);
			const V2 pn = tmpl_xf(xf, vt[j + 1 < N ? j + 1 : 0u]);
			const V2 pp = tmpl_xf(xf, vt[j > 0 ? j - 1 : N - 1]);
			mc.j = j;
			const Elem e = elem_geometry(mc, p1, v2dir(pp, p1), v2dir(p1, pn));
			nv = e.nv; ni = elem_total_indices(mc, e); inner = e.leftInner;
		}
		unsigned long long v = nv, i = ni;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const unsigned long long tv = __shfl_up(v, d), ti = __shfl_up(i, d);
			if (lane >= (uint32_t)d) { v += tv; i += ti; }
		}
		uint32_t nvP = __shfl_up(nv, 1);
		int inP = __shfl_up((int)inner, 1);
		if (lane == 63u) { s_wv[wave] = v; s_wi[wave] = i; s_wlastNv[wave] = nv; s_wlastIn[wave] = inner ? 1u : 0u; }
		__syncthreads();
		unsigned long long baseV = s_runV, baseI = s_runI, totV = 0, totI = 0;
#pragma unroll
		for (uint32_t w = 0; w < 4; ++w) { if (w < wave) { baseV += s_wv[w]; baseI += s_wi[w]; } totV += s_wv[w]; totI += s_wi[w]; }
		if (lane == 0) { nvP = wave > 0 ? s_wlastNv[wave - 1] : s_lastNv; inP = (int)(wave > 0 ? s_wlastIn[wave - 1] : s_lastIn); }
		if (j < N) { out[j] = tmpl_round_word((uint32_t)(baseV + v - nv), (uint32_t)(baseI + i - ni), nvP, inP != 0); }
		__syncthreads();
		const uint32_t lastJ = (N - j0 < 256u ? N - j0 : 256u) - 1u; // the chunk's last element
		if (tid == lastJ) { s_runV += totV; s_runI += totI; s_lastNv = nv; s_lastIn = inner ? 1u : 0u; }
		__syncthreads();
	}
	if (tid == 0) {
		A.rsz[2 * g] = s_runV; A.rsz[2 * g + 1] = s_runI;
		if (mc.closed) { out[0] = tmpl_round_word(0u, 0u, s_lastNv, s_lastIn != 0); }
	}
}

// After the sizes: ONE scan over every mesh of every instance (instance-major, the batch's mesh order): its first vertex / index in the
// batch (the template's sizes for the meshes without Round joins, the counted ones for the others; a mesh beyond 65 536 vertices ends the
// call like the ordinary path's scan, vgx_scan_ops.h), the batch totals, the caller's capacities. Parallel whatever the shape of the batch
// (ten thousand instances of a few hundred meshes, or one instance -- a static batch -- of millions).
struct OpTmplRoundMeshes
{
	VgxTmplArgs A;
	__device__ uint64_t size() const { return A.ninst * A.inst.num_meshes; }
	__device__ void sizes(uint64_t k, uint32_t* nv, uint32_t* ni, bool* tooLarge) const
	{
		const uint64_t M = A.inst.num_meshes;
		const uint64_t inst = k / M;
		const uint32_t m = (uint32_t)(k - inst * M);
		const uint2 ts = A.tmsz[m];
		*tooLarge = false;
		if (ts.x >> 31) {
			const unsigned long long* z = A.rsz + (inst * A.num_round + (ts.x & 0x7FFFFFFFu)) * 2;
			const unsigned long long v = z[0], i = z[1];
			*tooLarge = v > 65536ull; // what OpMeshOffsets reports for such a mesh (16-bit indices)
			*nv = vgx_sat_nv(v); *ni = vgx_sat_ni(i);
		} else {
			*nv = ts.x; *ni = ts.y;
		}
	}
	__device__ Sum3 load(uint64_t k) const
	{
		uint32_t nv, ni; bool big;
		sizes(k, &nv, &ni, &big);
		Sum3 r = sum3_zero(); r.a = nv; r.b = ni; r.c = big ? 1u : 0u;
		return r;
	}
	__device__ void store(uint64_t k, Sum3 e) const
	{
		uint32_t nv, ni; bool big;
		sizes(k, &nv, &ni, &big);
		VgxTmplMeshPlace q; q.v = e.a; q.i = e.b; q.nv = nv; q.ni = ni;
		A.mplace[k] = q;
	}
	__device__ void finish(Sum3 tot) const
	{
		vgx_sizes z = A.total;
		z.num_vertices = tot.a; z.num_indices = tot.b;
		A.totals->sizes = z;
		if (tot.c) { set_status(A.totals, VGX_E_MESH_TOO_LARGE); }
		const uint32_t aux = (tot.a > A.caps.vertices ? 1u : 0u) | (tot.b > A.caps.indices ? 2u : 0u) | ((A.meshes_out && z.num_meshes > A.caps.meshes) ? 4u : 0u);
		if (aux && A.totals->status == VGX_OK) { // the need is in sizes; nothing is emitted (the emit kernels leave when the status is set)
			A.totals->status = VGX_E_NOSPACE;
			A.totals->fail_reason = VGX_FAIL_OUT_CAPACITY;
			A.totals->fail_aux = aux;
		}
	}
};

// (k_tmpl_round_sizes_inst placed the meshes inside their instances:) the instances' places in the batch, the batch totals, the capacities
struct OpTmplRoundInst
{
	VgxTmplArgs A;
	__device__ uint64_t size() const { return A.ninst; }
	__device__ Sum3 load(uint64_t k) const { Sum3 r = sum3_zero(); r.a = A.itot[2 * k]; r.b = A.itot[2 * k + 1]; return r; }
	__device__ void store(uint64_t k, Sum3 e) const { A.iplace[2 * k] = e.a; A.iplace[2 * k + 1] = e.b; }
	__device__ void finish(Sum3 tot) const
	{
		vgx_sizes z = A.total;
		z.num_vertices = tot.a; z.num_indices = tot.b;
		A.totals->sizes = z;
		const uint32_t aux = (tot.a > A.caps.vertices ? 1u : 0u) | (tot.b > A.caps.indices ? 2u : 0u) | ((A.meshes_out && z.num_meshes > A.caps.meshes) ? 4u : 0u);
		if (aux && A.totals->status == VGX_OK) {
			A.totals->status = VGX_E_NOSPACE;
			A.totals->fail_reason = VGX_FAIL_OUT_CAPACITY;
			A.totals->fail_aux = aux;
		}
	}
};

} // namespace

bool vgx_tmpl_round_per_instance(const VgxTmplArgs& a) // which shape vgx_launch_tmpl_round_sizes takes: the host sets a.iplace / a.itot for this one
{
	// (several classes: num_round / num_round_elems are the sums over the classes -- the mean mesh length is the template's -- and round_lds the largest class)
	return a.num_round != 0 && a.num_round_elems / a.num_round <= 128u && (a.cls ? a.round_lds : a.num_round) <= VGX_TMPL_ROUND_MAXR && a.ninst >= 64;
}

void vgx_launch_tmpl_round_sizes(const VgxTmplArgs& a, Sum3* partial, hipStream_t s)
{
	const uint64_t blocks = a.ninst * a.tiles_per_inst; // the host checked < 2^31
	if (!blocks || a.num_round == 0) { return; } // (no Round-join mesh in the template: nothing to size, and no division by zero below)
	const uint64_t pairs = a.ninst * (uint64_t)a.num_round; // the host checked < 2^31
	if (a.num_round_elems / a.num_round > 128u) { hipLaunchKernelGGL(k_tmpl_round_sizes_block, dim3((unsigned)pairs), dim3(256), 0, s, a); } // long meshes: a workgroup each
	else if (vgx_tmpl_round_per_instance(a)) { // a workgroup per instance: sizes, the meshes' places inside the instance; then the scan over the instances
		hipLaunchKernelGGL(k_tmpl_round_sizes_inst, dim3((unsigned)a.ninst), dim3(256), (a.cls ? a.round_lds : a.num_round) * sizeof(TmplRoundRec), s, a);
		OpTmplRoundInst opi;
		opi.A = a;
		vgx_device_scan(opi, partial, s, a.ninst);
		return;
	} else { hipLaunchKernelGGL(k_tmpl_round_sizes, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, a); }
	OpTmplRoundMeshes op;
	op.A = a;
	vgx_device_scan(op, partial, s, a.ninst * a.inst.num_meshes);
}

void vgx_launch_tmpl_check(const vgx_draw* draws, uint64_t ndraws, uint32_t npaths, VgxTotals* totals, hipStream_t s)
{
	hipLaunchKernelGGL(k_tmpl_check, dim3(1024), dim3(256), 0, s, draws, ndraws, npaths, totals);
}

void vgx_launch_tmpl_hash(const vgx_draw* draws, uint64_t ndraws, uint64_t period, unsigned long long* hashes, hipStream_t s)
{
	hipLaunchKernelGGL(k_tmpl_hash, dim3(1024), dim3(256), 0, s, draws, ndraws, period, hashes);
}

void vgx_launch_tmpl_check_cls(const vgx_draw* draws, uint64_t ndraws, uint64_t period, const uint32_t* inst_cls, const uint32_t* cls_rep, VgxTotals* totals, hipStream_t s)
{
	hipLaunchKernelGGL(k_tmpl_check_cls, dim3(1024), dim3(256), 0, s, draws, ndraws, period, inst_cls, cls_rep, totals);
}

void vgx_launch_tmpl_styles(const VgxTmplBuild& b, hipStream_t s)
{
	const uint64_t gm = (b.num_meshes + 255) / 256;
	if (b.num_meshes) { hipLaunchKernelGGL(k_tmpl_styles, dim3((unsigned)(gm > 1024 ? 1024 : gm)), dim3(256), 0, s, b); }
}

void vgx_launch_tmpl_classes(const VgxTmplBuild& b, hipStream_t s)
{
	hipLaunchKernelGGL(k_tmpl_classes, dim3(1), dim3(1), 0, s, b);
}

void vgx_launch_tmpl_class_sums(const VgxTmplBuild& b, const vgx_draw_info* dinfo, const uint64_t* cmdPrefix, uint64_t numDraws, const vgx_sizes& all, unsigned long long* sums, hipStream_t s)
{
	hipLaunchKernelGGL(k_tmpl_class_sums, dim3(b.nclasses + 1), dim3(256), 0, s, b, dinfo, cmdPrefix, numDraws, all, sums);
}

void vgx_launch_tmpl_build(const VgxTmplBuild& b, hipStream_t s)
{
	const uint64_t gm = (b.num_meshes + 255) / 256;
	const uint64_t nt = (b.num_elems + b.tile - 1) / b.tile + b.nclasses; // >= the tile count (every class rounds up on its own)
	if (b.num_meshes) {
		hipLaunchKernelGGL(k_tmpl_meshes, dim3((unsigned)(gm > 4096 ? 4096 : gm)), dim3(256), 0, s, b);
		if (b.has_round) {
			OpTmplRoundIndex op; op.B = b; vgx_device_scan(op, b.partial, s, b.num_meshes);
			if (b.nclasses > 1) { hipLaunchKernelGGL(k_tmpl_round_classes, dim3(1), dim3(1), 0, s, b); }
		}
	}
	if (b.num_elems) {
		hipLaunchKernelGGL(k_tmpl_elems, dim3((unsigned)(nt > 65536 ? 65536 : nt)), dim3(256), 0, s, b); // one workgroup per tile
		hipLaunchKernelGGL(k_tmpl_tiles, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, b);
	}
}

void vgx_launch_tmpl_mtab(const VgxTmplArgs& a, vgx_mesh* mtab, VgxMeshDesc* mdesc, hipStream_t s)
{
	hipLaunchKernelGGL(k_tmpl_mtab, dim3(2048), dim3(256), 0, s, a, mtab, mdesc);
}

void vgx_launch_tmpl_emit(const VgxTmplArgs& a, hipStream_t s)
{
	const uint64_t blocks = a.wg ? a.num_wg : a.ninst * a.tiles_per_inst; // the host checked < 2^31
	if (!blocks) { return; }
	if (a.general == 6) { hipLaunchKernelGGL(k_tmpl_emit_round_aa_open, dim3((unsigned)blocks), dim3(VGX_TMPL_RC_THREADS), 0, s, a); }
	else if (a.general == 5) { hipLaunchKernelGGL(k_tmpl_emit_round_aa, dim3((unsigned)blocks), dim3(VGX_TMPL_RC_THREADS), 0, s, a); }
	else if (a.general == 4) { hipLaunchKernelGGL(k_tmpl_emit_bevel, dim3((unsigned)blocks), dim3(VGX_TMPL_THREADS), 0, s, a); }
	else if (a.general == 3) { hipLaunchKernelGGL(k_tmpl_emit_round, dim3((unsigned)blocks), dim3(VGX_TMPL_G_THREADS), 0, s, a); }
	else if (a.general == 2) { hipLaunchKernelGGL(k_tmpl_emit_general, dim3((unsigned)blocks), dim3(VGX_TMPL_G_THREADS), 0, s, a); }
	else if (a.general == 1) { hipLaunchKernelGGL(k_tmpl_emit_open, dim3((unsigned)blocks), dim3(VGX_TMPL_THREADS), 0, s, a); }
	else { hipLaunchKernelGGL(k_tmpl_emit, dim3((unsigned)blocks), dim3(VGX_TMPL_THREADS), 0, s, a); }
}
