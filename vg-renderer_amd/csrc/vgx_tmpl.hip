// vgx_tmpl.hip -- TEMPLATE mode of vgx_tessellate: a drawing submitted for many instances that differ only in their
// transform and colours (the BASELINE headline: Tiger x 10 000).
//
// The reference flattens a path in its LOCAL space (pathXXX, src/path.cpp:62-784, with tessTol = tol / scale^2) and only
// then applies the state transform to the finished polyline (transformPath -> vgutil::batchTransformPositions,
// src/vg.cpp:4957-4975, src/vg_util.cpp:266-272) before the stroker sees it. Instances whose draw records agree in
// everything the flattener and the stroker's SIZES depend on -- path, fill / stroke flags, stroke width, scale,
// tolerance, fringe -- therefore share one local polyline, one set of sub-paths and one set of mesh sizes (no Round
// joins: their point count depends on the transformed geometry, stroker.cpp:1146, 1592), bit for bit. What differs per
// instance is transformPos2D of every polyline vertex (vg_util.h:24-28) and everything the stroker derives from the
// transformed vertices (directions, extrusion vectors, inner side of every join, fill orientation).
//
// So vgx_tessellate_count flattens the FIRST period of such a batch once, with the ordinary two-phase kernels and
// apply_transform = 0, and keeps the result as a template (a few hundred KB, L2 resident): local polyline, per-mesh
// records with closed-form output offsets inside one instance, and an element table in processing order. One step of
// vgx_tessellate is then
//   k_tmpl_verify   every draw record against its image in the saved first period (fields above, bit patterns) + the
//                   finiteness checks of the ordinary path; a mismatch ends the call with VGX_E_STALE
//   k_tmpl_emit     one lane per ELEMENT (polyline vertex of one mesh of one instance): three template vertices from
//                   L2, transformPos2D with the instance's matrix in registers, the stroker's per-element arithmetic
//                   (strokerConvexFillAA stroker.cpp:713-807; closed Miter polylineStrokeAA / AAThin :1524-1579,
//                   1970-1984, 2060-2110, 2295-2306), stores at closed-form addresses -- no polyline heap, no scans, no
//                   mesh descriptors in HBM. The elements of an instance are processed in tiles; inside a tile the fill
//                   elements come first, then the stroke elements, so a wave's chunks are (almost) pure and the two
//                   mesh kinds of a draw, which interleave in the output streams, are written by the same wave within
//                   a few chunks (the hole pattern DESIGN.md section 9 measured at 2.4 TB/s when two kernels wrote them
//                   milliseconds apart).
// Results are identical to the ordinary path's by construction and by test (tests/test_gpu_tmpl.py: VGX_TMPL=0 vs 1
// byte for byte, both against the reference).
#include "vgx_internal.h"
#include "vgx_wave.h"
#include "vgx_elem.h"
#include "vgx_scan_ops.h"

namespace {

// ---- count pass: is the batch a template batch? ---------------------------------------------------------------------
// Bit patterns of the fields the flattener and the mesh sizes depend on; colours, transform and state_key may differ.
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

// ---- count pass: the template's tables from the first period's ordinary count + emit --------------------------------
__global__ __launch_bounds__(256) void k_tmpl_meshes(VgxTmplBuild B)
{
	const uint64_t M = B.num_meshes;
	for (uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (uint64_t)gridDim.x * blockDim.x) {
		const VgxMeshDesc md = B.mdesc[m];
		const VgxMeshPrep pr = B.mprep[m];
		const vgx_mesh mt = B.mtab[m];
		VgxTmplMesh t;
		t.poly_first = (uint32_t)md.poly_first;
		t.n = md.poly_n;
		t.v_off = (uint32_t)mt.first_vertex;
		t.i_off = (uint32_t)mt.first_index;
		t.drawk = md.draw;
		t.kind = md.kind;
		if (VGX_MD_KIND(md.kind) >= VGX_MESH_STROKE) { t.f0 = pr.f0; t.f1 = pr.f1; } // hsw / hswAA (thin: fringe, fringe)
		else { t.f0 = B.draws[md.draw].fringe * 0.5f; t.f1 = 0.0f; }                // |aa| = fringe / 2 (stroker.cpp:723); the sign is per instance
		B.tmesh[m] = t;
		B.tmtab[m] = mt;
	}
}

// Element table in processing order: tiles of `tile` elements of the instance's output-ordered element stream; inside a
// tile the fill elements first, then the stroke elements (both in output order).
__global__ __launch_bounds__(256) void k_tmpl_elems(VgxTmplBuild B)
{
	const uint64_t M = B.num_meshes;
	const uint64_t E = B.num_elems;
	const uint64_t T = B.tile;
	// owner of output-ordered element x: last mesh with fillPrefix + strokePrefix <= x; returns (mesh, fill elements before x, stroke elements before x)
	auto locate = [&](uint64_t x, uint64_t* mesh, uint64_t* fBefore, uint64_t* sBefore, uint32_t* j, bool* isFill) {
		if (x >= E) { *mesh = M; *fBefore = B.prefix_fill[M]; *sBefore = B.prefix_stroke[M]; *j = 0; *isFill = false; return; }
		uint64_t lo = 0, hi = M;
		while (hi - lo > 1) {
			const uint64_t mid = (lo + hi) >> 1;
			if (B.prefix_fill[mid] + B.prefix_stroke[mid] <= x) { lo = mid; } else { hi = mid; }
		}
		// zero-length entries cannot occur (every mesh has >= 2 elements), so lo owns x
		const uint64_t pf = B.prefix_fill[lo], psk = B.prefix_stroke[lo];
		const uint32_t jj = (uint32_t)(x - (pf + psk));
		const bool f = VGX_MD_KIND(B.mdesc[lo].kind) < VGX_MESH_STROKE;
		*mesh = lo; *j = jj; *isFill = f;
		*fBefore = pf + (f ? jj : 0u);
		*sBefore = psk + (f ? 0u : jj);
	};
	for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t m, f, s, m0, f0, s0, m1, f1, s1;
		uint32_t j, j0, j1;
		bool isFill, d0, d1;
		locate(e, &m, &f, &s, &j, &isFill);
		const uint64_t x0 = e / T * T;
		locate(x0, &m0, &f0, &s0, &j0, &d0);
		locate(x0 + T, &m1, &f1, &s1, &j1, &d1);
		const uint64_t slot = x0 + (isFill ? f - f0 : (f1 - f0) + (s - s0));
		VgxTmplElem r;
		r.mesh = (uint32_t)m;
		r.jq = j | ((uint32_t)(e - x0) << 16); // j < 65536 (a mesh holds at most 65536 vertices), position in the tile < tile <= 65536
		B.telem[slot] = r;
		if (e == x0) { B.tile_mesh0[e / T] = (uint32_t)m; }
	}
}

// ---- step: verify ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tmpl_verify(VgxTmplArgs A)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		// totals of the batch = instances x template (the memset in front of this kernel zeroed them)
		vgx_sizes z;
		z.num_poly_vertices = A.ninst * A.inst.num_poly_vertices;
		z.num_subpaths = A.ninst * A.inst.num_subpaths;
		z.num_meshes = A.ninst * A.inst.num_meshes;
		z.num_vertices = A.ninst * A.inst.num_vertices;
		z.num_indices = A.ninst * A.inst.num_indices;
		z.num_serial_draws = A.ninst * A.inst.num_serial_draws;
		z.num_cmd_instances = A.ninst * A.inst.num_cmd_instances;
		z.num_elements = A.ninst * A.inst.num_elements;
		z.num_fill_elements = A.ninst * A.inst.num_fill_elements;
		z.num_drawcmds = 0;
		A.totals->sizes = z;
		if (z.num_vertices > A.caps.vertices || z.num_indices > A.caps.indices || (A.meshes_out && z.num_meshes > A.caps.meshes)) {
			set_status(A.totals, VGX_E_NOSPACE);
			A.totals->fail_reason = VGX_FAIL_OUT_CAPACITY;
			A.totals->fail_aux = (z.num_vertices > A.caps.vertices ? 1u : 0u) | (z.num_indices > A.caps.indices ? 2u : 0u) | ((A.meshes_out && z.num_meshes > A.caps.meshes) ? 4u : 0u);
		}
	}
	const uint64_t P = A.period;
	bool bad = false;
	uint32_t err = VGX_OK;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t k = i % P;
	const uint64_t kstep = stride % P;
	for (; i < A.ndraws; i += stride) {
		const uint4* q = (const uint4*)(A.draws + i);
		const uint4* t = (const uint4*)(A.tdraws + k);
		const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
		const uint4 t0 = t[0], t1 = t[1], t2 = t[2];
		bad = bad || !tmpl_same(q0, q1, q2, t0, t1, t2);
		const uint32_t e = tmpl_validate(q0, q1, q2, q3, A.npaths);
		if (e != VGX_OK && err == VGX_OK) { err = e; }
		k += kstep;
		if (k >= P) { k -= P; }
	}
	if (err != VGX_OK) { set_status(A.totals, err); }
	if (bad) { set_status(A.totals, VGX_E_STALE); }
}

// ---- step: emit ------------------------------------------------------------------------------------------------------
// One workgroup = one TILE of one instance: `tile` consecutive elements of the instance's output-ordered element stream, i.e.
// a contiguous piece of each output stream (~40 KB), a contiguous range of the template's meshes and of its polyline.
//   phase 0  one thread per mesh of the tile: template mesh record + the instance's draw record (transform, colour) -> a
//            64-byte record in LDS; convex AA fills: the orientation of the TRANSFORMED polygon's first triangle
//            (stroker.cpp:721-723), once per mesh instead of once per element
//   phase 1  one lane per element: its template vertex from L2, transformPos2D (vg_util.h:24-28) ONCE, parked in LDS at the
//            element's output-order position inside the tile ("the growing polyline staged in LDS")
//   phase 2  one lane per element: the neighbouring corners (j - 2, j - 1, j + 1; the wrap-around corners of closed shapes)
//            from LDS -- from L2 + transform for the handful of elements whose neighbour lies in another tile --, the
//            stroker's per-element arithmetic, stores at closed-form addresses.
// Per 64 elements the kernel issues two global loads (element record, vertex) beside its stores; everything shared by the
// elements of a mesh comes from LDS.
struct TmplXf { float m0, m1, m2, m3, m4, m5; };
__device__ __forceinline__ V2 tmpl_xf(const TmplXf& m, float2 p) // transformPos2D, vg_util.h:24-28
{
	return v2(m.m0 * p.x + m.m2 * p.y + m.m4, m.m1 * p.x + m.m3 * p.y + m.m5);
}

// One element of a CLOSED stroke with MITER joins, AA (4 rails) or Thin (3 rails): stroke_chunk_simple (vgx_elem.h) without
// its neighbour lanes -- the previous join's inner side and, on the last element, join 0's are recomputed from the
// transformed vertices (vtx(jj) = transformed polyline vertex jj of the mesh) instead of being carried: same inputs, same
// arithmetic, same bits.
template<class VF>
__device__ __forceinline__ void tmpl_stroke_elem(uint32_t kindWord, uint32_t N, float hsw, float hswAA, uint32_t j, V2 p1, const VF& vtx, uint32_t color,
	float* posMesh, uint32_t* colMesh, uint16_t* idxMesh)
{
	const bool thin = VGX_MD_KIND(kindWord) == VGX_MESH_STROKE_AA_THIN;
	const uint32_t R = thin ? 3u : 4u;
	const uint32_t bridgeIdx = thin ? 12u : 18u;
	const float sideWidth = thin ? hsw : hswAA; // fringe : hswAA
	const uint32_t jn1 = j + 1 < N ? j + 1 : 0u;
	const uint32_t jp1 = j > 0 ? j - 1 : N - 1;
	const V2 pNext = vtx(jn1);
	const V2 pPrev = vtx(jp1);
	const V2 d12 = v2dir(p1, pNext);
	const V2 dPrev = v2dir(pPrev, p1);
	const VgxJoin jn = vgx_join_dirs(dPrev, d12, sideWidth);
	const bool L = jn.leftInner;
	const uint32_t b = R * j;
	const uint32_t top = b + R - 1;
	const Rails mine = thin ? (L ? rails(b, b + 1, b + 2, 0) : rails(top, b + 1, b, 0)) : (L ? rails(b, b + 1, b + 2, b + 3) : rails(top, b + 2, b + 1, b));
	const uint32_t c0 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
	float* pp = posMesh + 2 * (size_t)b;
	uint32_t* pc = colMesh + b;
	if (thin) { // stroker.cpp:2060-2110
		const V2 vf = v2mul(jn.v, hsw);
		const V2 q0 = L ? v2add(p1, vf) : v2sub(p1, vf);
		const V2 q2 = L ? v2sub(p1, vf) : v2add(p1, vf);
		PosPair q; q.x0 = q0.x; q.y0 = q0.y; q.x1 = p1.x; q.y1 = p1.y;
		VGX_ST_GUARD(c0) {
		*(PosPair*)pp = q;
		*(float2*)(pp + 4) = make_float2(q2.x, q2.y);
		ColPair c; c.c0 = c0; c.c1 = color;
		*(ColPair*)pc = c;
		pc[2] = c0;
		}
	} else { // :1524-1579
		const V2 vhaa = v2mul(jn.v, hswAA);
		const V2 vh = v2mul(jn.v, hsw);
		const V2 q0 = L ? v2add(p1, vhaa) : v2sub(p1, vhaa);
		const V2 q1 = L ? v2add(p1, vh) : v2sub(p1, vh);
		const V2 q2 = L ? v2sub(p1, vh) : v2add(p1, vh);
		const V2 q3 = L ? v2sub(p1, vhaa) : v2add(p1, vhaa);
		PosPair q; q.x0 = q0.x; q.y0 = q0.y; q.x1 = q1.x; q.y1 = q1.y;
		PosPair r; r.x0 = q2.x; r.y0 = q2.y; r.x1 = q3.x; r.y1 = q3.y;
		VGX_ST_GUARD(c0 ^ __float_as_uint(q.x0) ^ __float_as_uint(r.y1)) {
		*(PosPair*)pp = q;
		*(PosPair*)(pp + 4) = r;
		ColPair c; c.c0 = c0; c.c1 = color;
		*(ColPair*)pc = c;
		ColPair d; d.c0 = color; d.c1 = c0;
		*(ColPair*)(pc + 2) = d;
		}
	}
	if (j > 0) { // the bridge from the previous join (stroker.cpp:1557-1564, 1714-1721; thin :2093-2098, 2175-2180)
		const uint32_t jp2 = jp1 > 0 ? jp1 - 1 : N - 1;
		const V2 pPrev2 = vtx(jp2);
		const VgxJoin jp = vgx_join_dirs(v2dir(pPrev2, pPrev), dPrev, sideWidth);
		const uint32_t pb = R * (j - 1), ptop = pb + R - 1;
		const Rails p = thin ? (jp.leftInner ? rails(pb, pb + 1, pb + 2, 0) : rails(ptop, pb + 1, pb, 0))
		                     : (jp.leftInner ? rails(pb, pb + 1, pb + 2, pb + 3) : rails(ptop, pb + 2, pb + 1, pb));
		uint16_t* pi = idxMesh + (size_t)bridgeIdx * (j - 1);
		Idx6 t0; t0.a = (p.a & 0xFFFFu) | (p.b << 16); t0.b = (mine.b & 0xFFFFu) | (p.a << 16); t0.c = (mine.b & 0xFFFFu) | (mine.a << 16);
		Idx6 t1; t1.a = (p.b & 0xFFFFu) | (p.c << 16); t1.b = (mine.c & 0xFFFFu) | (p.b << 16); t1.c = (mine.c & 0xFFFFu) | (mine.b << 16);
		VGX_ST_GUARD(t0.a ^ t1.c) {
		*(Idx6*)pi = t0;
		*(Idx6*)(pi + 6) = t1;
		if (!thin) {
			Idx6 t2; t2.a = (p.c & 0xFFFFu) | (p.d << 16); t2.b = (mine.d & 0xFFFFu) | (p.c << 16); t2.c = (mine.d & 0xFFFFu) | (mine.c << 16);
			*(Idx6*)(pi + 12) = t2;
		}
		}
	}
	if (j + 1 == N) { // closing bridge to join 0 (:1970-1984, 2295-2306); d12 = vec2Dir(last vertex, vertex 0)
		const V2 v1 = vtx(N > 1 ? 1u : 0u);
		const VgxJoin j0 = vgx_join_dirs(d12, v2dir(pNext, v1), sideWidth);
		const Rails f = thin ? (j0.leftInner ? rails(0, 1, 2, 0) : rails(2, 1, 0, 0)) : (j0.leftInner ? rails(0, 1, 2, 3) : rails(3, 2, 1, 0));
		uint16_t* pi = idxMesh + (size_t)bridgeIdx * (N - 1);
		Idx6 t0; t0.a = (mine.a & 0xFFFFu) | (mine.b << 16); t0.b = (f.b & 0xFFFFu) | (mine.a << 16); t0.c = (f.b & 0xFFFFu) | (f.a << 16);
		Idx6 t1; t1.a = (mine.b & 0xFFFFu) | (mine.c << 16); t1.b = (f.c & 0xFFFFu) | (mine.b << 16); t1.c = (f.c & 0xFFFFu) | (f.b << 16);
		*(Idx6*)pi = t0;
		*(Idx6*)(pi + 6) = t1;
		if (!thin) {
			Idx6 t2; t2.a = (mine.c & 0xFFFFu) | (mine.d << 16); t2.b = (f.d & 0xFFFFu) | (mine.c << 16); t2.c = (f.d & 0xFFFFu) | (f.c << 16);
			*(Idx6*)(pi + 12) = t2;
		}
	}
}

// Per-mesh record of a tile in LDS (phase 0). 64 bytes.
struct __attribute__((aligned(16))) TmplRec
{
	uint32_t poly_first, n, v_off, i_off;
	uint32_t kind, color; float f0, f1; // fills: f0 = aa WITH the instance's orientation sign; strokes: hsw, hswAA
	float m0, m1, m2, m3;
	float m4, m5; uint32_t pad0, pad1;
};

// The per-mesh values of one (instance, template mesh): what phase 0 parks in LDS and what the fallback computes per lane.
__device__ __forceinline__ TmplRec tmpl_make_rec(const VgxTmplArgs& A, const vgx_draw* idraws, uint32_t m)
{
	const VgxTmplMesh tm = A.tmesh[m];
	const uint4* dq = (const uint4*)(idraws + tm.drawk);
	const uint32_t kind = VGX_MD_KIND(tm.kind);
	const bool isFill = kind < VGX_MESH_STROKE;
	const uint4 qc = dq[isFill ? 0 : 1]; // fill_color = q0.z, stroke_color = q1.x
	const uint4 q2 = dq[2], q3 = dq[3];
	TmplRec r;
	r.poly_first = tm.poly_first; r.n = tm.n; r.v_off = tm.v_off; r.i_off = tm.i_off;
	r.kind = tm.kind; r.color = isFill ? qc.z : qc.x; r.f0 = tm.f0; r.f1 = tm.f1;
	r.m0 = __uint_as_float(q2.y); r.m1 = __uint_as_float(q2.z); r.m2 = __uint_as_float(q2.w);
	r.m3 = __uint_as_float(q3.x); r.m4 = __uint_as_float(q3.y); r.m5 = __uint_as_float(q3.z);
	r.pad0 = 0; r.pad1 = 0;
	if (kind == VGX_MESH_FILL_AA) {
		// orientation from the first triangle of the TRANSFORMED polygon (stroker.cpp:721-723)
		TmplXf xf; xf.m0 = r.m0; xf.m1 = r.m1; xf.m2 = r.m2; xf.m3 = r.m3; xf.m4 = r.m4; xf.m5 = r.m5;
		const float2* vt = A.tpoly + tm.poly_first;
		const V2 a0 = tmpl_xf(xf, vt[0]), a1 = tmpl_xf(xf, vt[1]), a2 = tmpl_xf(xf, vt[2]);
		const float orient = v2cross(v2sub(a1, a0), v2sub(a2, a0));
		r.f0 = tm.f0 * vgm_sign(orient);
	}
	return r;
}

// One element once its mesh record, its own transformed vertex and a way to get the mesh's other transformed vertices exist.
template<class VF>
__device__ __forceinline__ void tmpl_elem_emit(const VgxTmplArgs& A, uint64_t inst, uint32_t mesh, uint32_t j, const TmplRec& r, V2 p1, const VF& vtx)
{
	const uint32_t kind = VGX_MD_KIND(r.kind);
	const uint32_t N = r.n;
	if (j == 0 && A.meshes_out) { // the caller's mesh table: the template's record moved to this instance
		vgx_mesh mr = A.tmtab[mesh];
		mr.first_vertex += inst * A.inst.num_vertices;
		mr.first_index += inst * A.inst.num_indices;
		mr.draw += (uint32_t)(inst * A.period);
		A.meshes_out[inst * A.inst.num_meshes + mesh] = mr;
	}
	if (kind < VGX_MESH_STROKE) {
		FillFetch F;
		F.valid = true; F.j = j; F.N = N; F.color = r.color; F.ibase = 0; F.mi = 0;
		F.firstV = inst * A.inst.num_vertices + r.v_off;
		F.firstI = inst * A.inst.num_indices + r.i_off;
		F.aaElem = kind == VGX_MESH_FILL_AA;
		F.sseOrder = VGX_MD_SSE_ORDER(r.kind) != 0;
		F.nextInWave = false; F.prevInWave = false;
		F.p1 = p1; F.pNextB = p1; F.pPrevB = p1;
		F.aa = r.f0;
		V2 dPrev = v2(0.0f, 0.0f), d12 = dPrev;
		if (F.aaElem) {
			d12 = v2dir(p1, vtx(j + 1 < N ? j + 1 : 0u));
			dPrev = v2dir(vtx(j > 0 ? j - 1 : N - 1), p1);
		}
		fill_emit_store(A.pos, A.color, A.idx, F, dPrev, d12);
	} else {
		const uint64_t v0 = inst * A.inst.num_vertices + r.v_off;
		tmpl_stroke_elem(r.kind, N, r.f0, r.f1, j, p1, vtx, r.color, A.pos + 2 * v0, A.color + v0, A.idx + (inst * A.inst.num_indices + r.i_off));
	}
}

#define VGX_TMPL_THREADS 256
#define VGX_TMPL_MAX_TILE 1024 /* elements per tile the LDS vertex stage holds */
#define VGX_TMPL_MAXM 96       /* meshes per tile the LDS record table holds; a tile that touches more takes the per-lane fallback */
#define VGX_TMPL_CH (VGX_TMPL_MAX_TILE / VGX_TMPL_THREADS)
#ifndef VGX_TMPL_OCC
#define VGX_TMPL_OCC
#endif
__global__ __launch_bounds__(VGX_TMPL_THREADS) VGX_TMPL_OCC void k_tmpl_emit(VgxTmplArgs A)
{
	__shared__ TmplRec s_rec[VGX_TMPL_MAXM];
	__shared__ float2 s_vtx[VGX_TMPL_MAX_TILE];
	if (A.totals->status != VGX_OK) {
		return;
	}
	const uint32_t tid = threadIdx.x;
	const uint32_t inst32 = blockIdx.x / A.tiles_per_inst;
	const uint32_t t = blockIdx.x - inst32 * A.tiles_per_inst;
	const uint64_t inst = inst32;
	const uint32_t E = (uint32_t)A.inst.num_elements;
	const uint32_t x0 = t * A.tile;
	const uint32_t nel = x0 + A.tile < E ? A.tile : E - x0;
	const uint32_t mA = A.tile_mesh0[t];
	const uint32_t mB = t + 1 < A.tiles_per_inst ? A.tile_mesh0[t + 1] : (uint32_t)A.inst.num_meshes - 1;
	const uint32_t nm = mB - mA + 1;
	const vgx_draw* idraws = A.draws + inst * A.period;
	const VgxTmplElem* telem = A.telem + x0;
	if (nm > VGX_TMPL_MAXM) { // block-uniform. Many tiny meshes in one tile: every lane fetches its own records
		for (uint32_t s = tid; s < nel; s += VGX_TMPL_THREADS) {
			const VgxTmplElem er = telem[s];
			const TmplRec r = tmpl_make_rec(A, idraws, er.mesh);
			TmplXf xf; xf.m0 = r.m0; xf.m1 = r.m1; xf.m2 = r.m2; xf.m3 = r.m3; xf.m4 = r.m4; xf.m5 = r.m5;
			const float2* vt = A.tpoly + r.poly_first;
			const uint32_t j = er.jq & 0xFFFFu;
			tmpl_elem_emit(A, inst, er.mesh, j, r, tmpl_xf(xf, vt[j]), [&](uint32_t jj) { return tmpl_xf(xf, vt[jj]); });
		}
		return;
	}
	// phase 0
	if (tid < nm) { s_rec[tid] = tmpl_make_rec(A, idraws, mA + tid); }
	__syncthreads();
	// phase 1: chunk k = c * 4 + wave of the tile (interleaved: the tile's stroke chunks, the heavier ones, spread over the waves)
	VgxTmplElem er[VGX_TMPL_CH];
	V2 p1[VGX_TMPL_CH];
#pragma unroll
	for (int c = 0; c < VGX_TMPL_CH; ++c) {
		const uint32_t s = (uint32_t)c * VGX_TMPL_THREADS + tid;
		er[c].mesh = mA; er[c].jq = 0;
		if (s < nel) { er[c] = telem[s]; }
	}
#pragma unroll
	for (int c = 0; c < VGX_TMPL_CH; ++c) {
		const uint32_t s = (uint32_t)c * VGX_TMPL_THREADS + tid;
		p1[c] = v2(0.0f, 0.0f);
		if (s < nel) {
			const TmplRec* r = &s_rec[er[c].mesh - mA];
			const float2 lp = A.tpoly[r->poly_first + (er[c].jq & 0xFFFFu)];
			TmplXf xf; xf.m0 = r->m0; xf.m1 = r->m1; xf.m2 = r->m2; xf.m3 = r->m3; xf.m4 = r->m4; xf.m5 = r->m5;
			p1[c] = tmpl_xf(xf, lp);
			s_vtx[er[c].jq >> 16] = make_float2(p1[c].x, p1[c].y);
		}
	}
	__syncthreads();
	// phase 2
#pragma unroll
	for (int c = 0; c < VGX_TMPL_CH; ++c) {
		const uint32_t s = (uint32_t)c * VGX_TMPL_THREADS + tid;
		if (s < nel) {
			const TmplRec* rp = &s_rec[er[c].mesh - mA];
			TmplRec r;
			r.poly_first = rp->poly_first; r.n = rp->n; r.v_off = rp->v_off; r.i_off = rp->i_off;
			r.kind = rp->kind; r.color = rp->color; r.f0 = rp->f0; r.f1 = rp->f1;
			const uint32_t j = er[c].jq & 0xFFFFu;
			const int q0 = (int)(er[c].jq >> 16) - (int)j; // tile position of the mesh's vertex 0 (negative: in front of the tile)
			tmpl_elem_emit(A, inst, er[c].mesh, j, r, p1[c], [&](uint32_t jj) {
				const int qq = q0 + (int)jj;
				if (qq >= 0 && qq < (int)nel) { const float2 v = s_vtx[qq]; return v2(v.x, v.y); }
				TmplXf xf; xf.m0 = rp->m0; xf.m1 = rp->m1; xf.m2 = rp->m2; xf.m3 = rp->m3; xf.m4 = rp->m4; xf.m5 = rp->m5;
				return tmpl_xf(xf, A.tpoly[r.poly_first + jj]); // the neighbour belongs to another tile
			});
		}
	}
}

} // namespace

void vgx_launch_tmpl_check(const vgx_draw* draws, uint64_t ndraws, uint32_t npaths, VgxTotals* totals, hipStream_t s)
{
	hipLaunchKernelGGL(k_tmpl_check, dim3(1024), dim3(256), 0, s, draws, ndraws, npaths, totals);
}

void vgx_launch_tmpl_build(const VgxTmplBuild& b, hipStream_t s)
{
	const uint64_t gm = (b.num_meshes + 255) / 256, ge = (b.num_elems + 255) / 256;
	if (b.num_meshes) { hipLaunchKernelGGL(k_tmpl_meshes, dim3((unsigned)(gm > 4096 ? 4096 : gm)), dim3(256), 0, s, b); }
	if (b.num_elems) { hipLaunchKernelGGL(k_tmpl_elems, dim3((unsigned)(ge > 4096 ? 4096 : ge)), dim3(256), 0, s, b); }
}

void vgx_launch_tmpl_verify(const VgxTmplArgs& a, hipStream_t s)
{
	const uint64_t g = (a.ndraws + 255) / 256;
	hipLaunchKernelGGL(k_tmpl_verify, dim3((unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g))), dim3(256), 0, s, a);
}

void vgx_launch_tmpl_emit(const VgxTmplArgs& a, hipStream_t s)
{
	const uint64_t blocks = a.ninst * a.tiles_per_inst; // the host checked < 2^31
	if (blocks) { hipLaunchKernelGGL(k_tmpl_emit, dim3((unsigned)blocks), dim3(VGX_TMPL_THREADS), 0, s, a); }
}
