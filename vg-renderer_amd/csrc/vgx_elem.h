// vgx_elem.h -- one stroker ELEMENT per lane: geometry, counts, exit rails and the emitter of one polyline vertex of
// one mesh (convex-fill corner, stroke cap or join), used by the element kernels of vgx_stroke.hip. The vertex source
// is a template parameter (VS: `V2 ld(uint32_t i) const` returns vertex i of the mesh's polyline; VtxGlobal = the heap).
// Every emitted position / colour / index follows the cited reference lines (src/stroker.cpp); the rails
// formulation is the one of SURVEY.md appendix B.
#ifndef VGX_ELEM_H
#define VGX_ELEM_H

#include "vgx_internal.h"
#include "vgx_wave.h"

// Per-element functions (one lane = one polyline vertex of one mesh) are host + device: the kernels of vgx_stroke.hip /
// vgx_tmpl.hip run them one element per lane, the host backend of the per-call compat layer (host/vgx_host_backend.hip)
// runs the same functions element after element. The wave-level drivers (stroke_chunk*, fill_emit_chunk, round_mesh_size)
// are device only.
#define VGX_EL __host__ __device__ __forceinline__

namespace {

// Tuning builds only (profiles/ab_variants.sh): VGX_EXP_NOSTORE makes every output store of the element kernels conditional
// on a value that never occurs (the address arithmetic and the data stay live), VGX_EXP_NOLOAD replaces the polyline
// reads by synthetic vertices. Neither is ever defined in the product build.
#ifdef VGX_EXP_NOSTORE
#define VGX_ST_GUARD(x) if ((x) == 0x7FEDCBA9u)
#else
#define VGX_ST_GUARD(x)
#endif

struct Rails { uint32_t a, b, c, d; }; // AA: laa,l,r,raa   non-AA: l,r,-,-   thin: laa,m,raa,-

VGX_EL Rails rails(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { Rails r; r.a = a; r.b = b; r.c = c; r.d = d; return r; }
VGX_EL uint64_t rails_pack(Rails r) { return (uint64_t)(r.a & 0xFFFFu) | ((uint64_t)(r.b & 0xFFFFu) << 16) | ((uint64_t)(r.c & 0xFFFFu) << 32) | ((uint64_t)(r.d & 0xFFFFu) << 48); }
VGX_EL Rails rails_unpack(uint64_t p) { return rails((uint32_t)(p & 0xFFFFu), (uint32_t)((p >> 16) & 0xFFFFu), (uint32_t)((p >> 32) & 0xFFFFu), (uint32_t)(p >> 48)); }

// Unaligned wide stores: the output streams are only element-aligned (8 / 4 / 2 bytes); gfx950 global stores handle
// that natively (unaligned access mode), so one lane can issue ONE dwordx4 for two positions, ONE dwordx3 for six
// indices ... without alignment-dependent divergence.
struct __attribute__((packed, aligned(8))) PosPair { float x0, y0, x1, y1; };
struct __attribute__((packed, aligned(4))) ColPair { uint32_t c0, c1; };
struct __attribute__((packed, aligned(2))) Idx9 { uint32_t a, b, c, d; uint16_t e; };
struct __attribute__((packed, aligned(2))) Idx6 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(2))) Idx3 { uint32_t a; uint16_t b; };

// Stroke writer: pointers to the mesh's first vertex / index in the output streams, plus a register stage for the
// fixed-size part of an element (up to 4 vertices at b.., up to 24 indices at k..). The element code runs under
// divergent branches (cap / join / kind); without the stage every branch carries its own train of narrow stores
// (~60 store instructions per 64-element chunk), with it the chunk leaves in ~10 wide ones (flush()).
struct StrokeWriter
{
	float* pos;
	uint32_t* col;
	uint16_t* idx;
	uint32_t color, c0;
	uint32_t ib; // added to every index written (assembly: vertices in front of the mesh inside its vertex buffer)
	uint32_t vo = 0, io = 0; // subtracted from every vertex / index POSITION (not value): pos / col / idx may point at the LDS stage of ONE chunk, which begins at the chunk's first vertex / index (stroke_chunk)
	float sx[4], sy[4];
	uint32_t sc[4];
	uint32_t si[24];
	uint32_t nvS, niS; // staged vertex / index counts (niS is a multiple of 6)
	VGX_EL void reset()
	{
		for (int i = 0; i < 4; ++i) { sx[i] = 0.0f; sy[i] = 0.0f; sc[i] = 0; }
		for (int i = 0; i < 24; ++i) { si[i] = 0; }
		nvS = 0; niS = 0;
	}
	// ---- staged (slot = compile-time constant at every call site) ----
	VGX_EL void sv(uint32_t slot, V2 p, uint32_t c)
	{
		sx[slot] = p.x; sy[slot] = p.y; sc[slot] = c;
		nvS = nvS > slot + 1 ? nvS : slot + 1;
	}
	VGX_EL void stri(uint32_t slot, uint32_t a, uint32_t b, uint32_t c)
	{
		si[slot] = a; si[slot + 1] = b; si[slot + 2] = c;
		niS = niS > slot + 3 ? niS : slot + 3;
	}
	VGX_EL void sbridge4(uint32_t slot, Rails p, Rails c) // stroker.cpp:1557-1564, 1714-1721, 1973-1980
	{
		stri(slot, p.a, p.b, c.b); stri(slot + 3, p.a, c.b, c.a);
		stri(slot + 6, p.b, p.c, c.c); stri(slot + 9, p.b, c.c, c.b);
		stri(slot + 12, p.c, p.d, c.d); stri(slot + 15, p.c, c.d, c.c);
	}
	VGX_EL void sbridge2(uint32_t slot, Rails p, Rails c) // stroker.cpp:1119-1122, 1217-1220, 1374-1377
	{
		stri(slot, p.a, p.b, c.b); stri(slot + 3, p.a, c.b, c.a);
	}
	VGX_EL void sbridge3(uint32_t slot, Rails p, Rails c) // stroker.cpp:2093-2098, 2175-2180, 2299-2304
	{
		stri(slot, p.a, p.b, c.b); stri(slot + 3, p.a, c.b, c.a);
		stri(slot + 6, p.b, p.c, c.c); stri(slot + 9, p.b, c.c, c.b);
	}
	// ---- direct ----
	VGX_EL void v(uint32_t i, V2 p, uint32_t c) const
	{
		VGX_ST_GUARD(c) {
		*(float2*)(pos + 2 * (size_t)(i - vo)) = make_float2(p.x, p.y);
		col[i - vo] = c;
		}
	}
	// two consecutive vertices in one 16-byte + one 8-byte store (the arc loops of Round joins / caps write pairs)
	VGX_EL void v2(uint32_t i, V2 p, uint32_t c, V2 q, uint32_t d) const
	{
		VGX_ST_GUARD(c ^ d) {
		PosPair pp; pp.x0 = p.x; pp.y0 = p.y; pp.x1 = q.x; pp.y1 = q.y;
		*(PosPair*)(pos + 2 * (size_t)(i - vo)) = pp;
		ColPair cp; cp.c0 = c; cp.c1 = d;
		*(ColPair*)(col + (i - vo)) = cp;
		}
	}
	// three consecutive triangles (18 contiguous bytes) in one 16-byte + one 2-byte store
	VGX_EL void tri3(uint32_t k, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t c0, uint32_t c1, uint32_t c2) const
	{
		Idx9 t;
		t.a = ((a0 + ib) & 0xFFFFu) | ((a1 + ib) << 16); t.b = ((a2 + ib) & 0xFFFFu) | ((b0 + ib) << 16);
		t.c = ((b1 + ib) & 0xFFFFu) | ((b2 + ib) << 16); t.d = ((c0 + ib) & 0xFFFFu) | ((c1 + ib) << 16);
		t.e = (uint16_t)(c2 + ib);
		VGX_ST_GUARD(t.a ^ t.d) { *(Idx9*)(idx + (k - io)) = t; }
	}
	VGX_EL void tri(uint32_t k, uint32_t a, uint32_t b, uint32_t c) const
	{
		Idx3 t; t.a = ((a + ib) & 0xFFFFu) | ((b + ib) << 16); t.b = (uint16_t)(c + ib);
		VGX_ST_GUARD(t.a) { *(Idx3*)(idx + (k - io)) = t; }
	}
	VGX_EL void bridge4(uint32_t k, Rails p, Rails c) const
	{
		tri(k, p.a, p.b, c.b); tri(k + 3, p.a, c.b, c.a);
		tri(k + 6, p.b, p.c, c.c); tri(k + 9, p.b, c.c, c.b);
		tri(k + 12, p.c, p.d, c.d); tri(k + 15, p.c, c.d, c.c);
	}
	VGX_EL void bridge2(uint32_t k, Rails p, Rails c) const
	{
		tri(k, p.a, p.b, c.b); tri(k + 3, p.a, c.b, c.a);
	}
	VGX_EL void bridge3(uint32_t k, Rails p, Rails c) const
	{
		tri(k, p.a, p.b, c.b); tri(k + 3, p.a, c.b, c.a);
		tri(k + 6, p.b, p.c, c.c); tri(k + 9, p.b, c.c, c.b);
	}
	// the staged part of the element whose first vertex is b and first index k
	VGX_EL void flush(uint32_t b, uint32_t k) const
	{
		VGX_ST_GUARD(sc[0] ^ si[0] ^ si[7] ^ __float_as_uint(sx[3]) ^ __float_as_uint(sy[1]) ^ sc[3] ^ si[23] ^ si[13]) { flush_(b, k); }
	}
	VGX_EL void flush_(uint32_t b, uint32_t k) const
	{
		float* pp = pos + 2 * (size_t)(b - vo);
		uint32_t* pc = col + (b - vo);
		if (nvS >= 2) {
			PosPair q; q.x0 = sx[0]; q.y0 = sy[0]; q.x1 = sx[1]; q.y1 = sy[1];
			*(PosPair*)pp = q;
			ColPair c; c.c0 = sc[0]; c.c1 = sc[1];
			*(ColPair*)pc = c;
		}
		if (nvS == 4) {
			PosPair q; q.x0 = sx[2]; q.y0 = sy[2]; q.x1 = sx[3]; q.y1 = sy[3];
			*(PosPair*)(pp + 4) = q;
			ColPair c; c.c0 = sc[2]; c.c1 = sc[3];
			*(ColPair*)(pc + 2) = c;
		}
		if (nvS == 3) {
			*(float2*)(pp + 4) = make_float2(sx[2], sy[2]);
			pc[2] = sc[2];
		}
		uint16_t* pi = idx + (k - io);
#pragma unroll
		for (uint32_t g = 0; g < 4; ++g) {
			if (niS > 6 * g) {
				Idx6 t;
				t.a = ((si[6 * g] + ib) & 0xFFFFu) | ((si[6 * g + 1] + ib) << 16);
				t.b = ((si[6 * g + 2] + ib) & 0xFFFFu) | ((si[6 * g + 3] + ib) << 16);
				t.c = ((si[6 * g + 4] + ib) & 0xFFFFu) | ((si[6 * g + 5] + ib) << 16);
				*(Idx6*)(pi + 6 * g) = t;
			}
		}
	}
};

enum { ET_CAP_FIRST = 0, ET_JOIN = 1, ET_CAP_LAST = 2 };

// Everything step A computes for one stroke element and step D needs again.
struct Elem
{
	uint32_t nv, ni;  // my vertex / index count (ni excludes the connect / closing bridges)
	uint32_t et;
	V2 p1;            // the polyline vertex this element sits on
	V2 d01, d12, v;   // join: segment directions + extrusion; caps: d01 = cap direction
	bool leftInner;
	bool hasConnect;  // a bridge from the previous element precedes my own indices
	bool closesLoop;  // I am the last join of a closed stroke: closing bridge follows my indices
	VgxArc arc;       // Round/Bevel joins (n = 1 for Bevel)
	uint32_t H;       // numPointsHalfCircle (Round caps)
};

// Vertex source: the mesh's polyline in the heap (HBM).
struct VtxGlobal
{
	const float2* p;
#ifdef VGX_EXP_NOLOAD
	VGX_EL V2 ld(uint32_t i) const { const uint32_t h = (i + (uint32_t)(size_t)p) * 2654435761u; return v2((float)(h & 1023u), (float)((h >> 10) & 1023u)); }
#else
	VGX_EL V2 ld(uint32_t i) const { const float2 t = p[i]; return v2(t.x, t.y); }
#endif
};
template<class VS>
struct MeshCtxT
{
	uint32_t kind, N, j, cap, join;
	bool closed;
	float hsw, hswAA, fringe;
	const vgx_draw* dr; // scale / tolerance are only read where Round caps / joins need da
	VS vtx;             // the mesh's polyline
	float da = -1.0f;   // the mesh's arc step when the caller has it already (template kernels: once per mesh, with the template); else < 0: mesh_da evaluates it
};
typedef MeshCtxT<VtxGlobal> MeshCtx;

VGX_EL V2 ldv(const float* vtx, uint32_t i)
{
#ifdef VGX_EXP_NOLOAD
	{ const uint32_t h = (i + (uint32_t)(size_t)vtx) * 2654435761u; return v2((float)(h & 1023u), (float)((h >> 10) & 1023u)); }
#endif
	const float2 t = *(const float2*)(vtx + 2 * (size_t)i);
	return v2(t.x, t.y);
}

template<class VS>
VGX_EL float mesh_da(const MeshCtxT<VS>& m) // stroker.cpp:1013, 1398 (da uses hsw WITHOUT the fringe)
{
	if (m.da >= 0.0f) { return m.da; } // (the same function of the same three values, evaluated by whoever filled it in)
	return vgx_step_angle(m.dr->scale, m.hsw, m.dr->tess_tol);
}

VGX_EL MeshCtx make_mesh_ctx(const VgxMeshDesc& md, const VgxMeshPrep& pr, const vgx_draw* draws, uint32_t j, const float* poly)
{
	MeshCtx mc;
	mc.kind = VGX_MD_KIND(md.kind);
	mc.closed = VGX_MD_CLOSED(md.kind) != 0;
	mc.cap = VGX_MD_CAP(md.kind);
	mc.join = VGX_MD_JOIN(md.kind);
	mc.N = md.poly_n;
	mc.j = j;
	mc.vtx.p = (const float2*)poly + md.poly_first;
	mc.hsw = pr.f0; mc.hswAA = pr.f1; mc.fringe = pr.f2;
	mc.dr = draws + md.draw;
	return mc;
}

// ---- step A (strokes) --------------------------------------------------------------------------------
// p1 = the element's polyline vertex, dPrev = vec2Dir(previous vertex, p1), d12 = vec2Dir(p1, next vertex) (cyclic).
template<class VS>
VGX_EL Elem elem_geometry(const MeshCtxT<VS>& m, V2 p1, V2 dPrev, V2 d12)
{
	Elem e;
	e.nv = 0; e.ni = 0; e.et = ET_JOIN; e.leftInner = true; e.hasConnect = false; e.closesLoop = false;
	e.arc.a01 = 0.0f; e.arc.arcDa = 0.0f; e.arc.n = 1; e.H = 2;
	const uint32_t N = m.N, j = m.j;
	e.p1 = p1;
	e.d01 = v2(0.0f, 0.0f); e.d12 = e.d01; e.v = e.d01;
	// polyline strokes
	const bool isCapFirst = !m.closed && j == 0;
	const bool isCapLast = !m.closed && j == N - 1;
	const uint32_t railCount = (m.kind == VGX_MESH_STROKE) ? 2u : (m.kind == VGX_MESH_STROKE_AA ? 4u : 3u);
	const uint32_t bridgeIdx = (railCount - 1) * 6; // 6 / 18 / 12
	if (isCapFirst || isCapLast) {
		e.et = isCapFirst ? ET_CAP_FIRST : ET_CAP_LAST;
		e.d01 = isCapFirst ? d12 : dPrev;
		e.hasConnect = isCapLast;
		const bool roundCap = (m.cap == VGX_CAP_ROUND) && m.kind != VGX_MESH_STROKE_AA_THIN;
		if (roundCap) {
			const uint32_t H = vgx_half_circle_points(mesh_da(m));
			e.H = H;
			if (m.kind == VGX_MESH_STROKE_AA) {
				e.nv = 2 * H;
				e.ni = isCapFirst ? (9 * H - 12) : (3 * (H - 2) + 6 * (H - 1)); // stroker.cpp:1490, 1949-1966
			} else {
				e.nv = H;
				e.ni = 3 * (H - 2); // stroker.cpp:1071, 1362
			}
		} else {
			e.nv = railCount;
			e.ni = (m.kind == VGX_MESH_STROKE_AA) ? 6u : 0u; // cap quad only exists in the AA stroker
		}
		return e;
	}
	e.et = ET_JOIN;
	// (three selects on VALUES: `kind == A ? m.hsw : (kind == B ? m.hswAA : m.fringe)` was compiled into an indexed load from the
	// context structure, which then had to live in scratch memory wherever it was built field by field)
	const float wS = m.hsw, wA = m.hswAA, wT = m.fringe;
	float sideWidth = wT;
	sideWidth = m.kind == VGX_MESH_STROKE_AA ? wA : sideWidth;
	sideWidth = m.kind == VGX_MESH_STROKE ? wS : sideWidth;
	const VgxJoin jn = vgx_join_dirs(dPrev, d12, sideWidth);
	e.d01 = jn.d01; e.d12 = jn.d12; e.v = jn.v; e.leftInner = jn.leftInner;
	e.hasConnect = !(m.closed && j == 0);
	e.closesLoop = m.closed && j == N - 1;
	if (m.kind == VGX_MESH_STROKE_AA_THIN) { // stroker.cpp:2060-2240; Round join -> Bevel (:318-327)
		const bool bevel = m.join != VGX_JOIN_MITER;
		e.nv = bevel ? 4 : 3;
		e.ni = bevel ? 3 : 0;
		return e;
	}
	if (m.join == VGX_JOIN_MITER) {
		e.nv = railCount;
		e.ni = 0;
		return e;
	}
	if (m.join == VGX_JOIN_ROUND) {
		const V2 n01 = e.leftInner ? v2cw(e.d01) : v2ccw(e.d01);
		const V2 n12 = e.leftInner ? v2cw(e.d12) : v2ccw(e.d12);
		e.arc = vgx_round_join_arc(n01, n12, e.leftInner, mesh_da(m));
	}
	const uint32_t n = e.arc.n;
	if (m.kind == VGX_MESH_STROKE_AA) {
		e.nv = 2 * n + 4; // stroker.cpp:1599
		e.ni = 9 * n;     // :1675
	} else {
		e.nv = n + 2;     // :1156
		e.ni = 3 * n;     // :1186
	}
	(void)bridgeIdx;
	return e;
}

template<class VS>
VGX_EL uint32_t elem_total_indices(const MeshCtxT<VS>& m, const Elem& e)
{
	const uint32_t bridgeIdx = (m.kind == VGX_MESH_STROKE) ? 6u : (m.kind == VGX_MESH_STROKE_AA ? 18u : 12u);
	return e.ni + (e.hasConnect ? bridgeIdx : 0u) + (e.closesLoop ? bridgeIdx : 0u);
}

// ---- exit rails (what the next element connects to) ---------------------------------------------------
template<class VS>
VGX_EL Rails elem_exit_rails(const MeshCtxT<VS>& m, const Elem& e, uint32_t b)
{
	if (m.kind == VGX_MESH_STROKE_AA) {
		if (e.et == ET_CAP_FIRST) {
			if (m.cap == VGX_CAP_ROUND) { return rails(1, 0, (e.H - 1) * 2, (e.H - 1) * 2 + 1); } // :1512-1515
			return rails(0, 1, 2, 3);
		}
		if (m.join == VGX_JOIN_MITER) {
			return e.leftInner ? rails(b, b + 1, b + 2, b + 3) : rails(b + 3, b + 2, b + 1, b);
		}
		const uint32_t arcID = b + 2 + 2 * e.arc.n;
		return e.leftInner ? rails(b, b + 1, arcID, arcID + 1) : rails(arcID + 1, arcID, b + 1, b);
	}
	if (m.kind == VGX_MESH_STROKE) {
		if (e.et == ET_CAP_FIRST) {
			if (m.cap == VGX_CAP_ROUND) { return rails(0, e.H - 1, 0, 0); } // :1079-1080
			return rails(0, 1, 0, 0);
		}
		if (m.join == VGX_JOIN_MITER) {
			return e.leftInner ? rails(b, b + 1, 0, 0) : rails(b + 1, b, 0, 0);
		}
		const uint32_t endID = b + e.arc.n + 1;
		return e.leftInner ? rails(b, endID, 0, 0) : rails(endID, b, 0, 0);
	}
	// thin
	if (e.et == ET_CAP_FIRST) {
		return rails(0, 1, 2, 0);
	}
	if (m.join == VGX_JOIN_MITER) {
		return e.leftInner ? rails(b, b + 1, b + 2, 0) : rails(b + 2, b + 1, b, 0);
	}
	return e.leftInner ? rails(b, b + 1, b + 3, 0) : rails(b + 3, b + 1, b, 0);
}

// entry rails of join 0 of a closed stroke = the firstSegment*ID of the reference
template<class VS>
VGX_EL Rails first_join_entry(const MeshCtxT<VS>& m, bool leftInner0)
{
	if (m.kind == VGX_MESH_STROKE_AA) { return leftInner0 ? rails(0, 1, 2, 3) : rails(3, 2, 1, 0); }
	if (m.kind == VGX_MESH_STROKE) { return leftInner0 ? rails(0, 1, 0, 0) : rails(1, 0, 0, 0); }
	return leftInner0 ? rails(0, 1, 2, 0) : rails(2, 1, 0, 0);
}

// ---- step D: emit one stroke element ------------------------------------------------------------------
// Fixed-size pieces (Butt / Square caps, the four corner vertices of a join, the connect bridge) go through the
// writer's register stage (sv / stri / sbridgeN, slot numbers relative to b / k) and leave in a handful of wide stores
// after the call; variable-size pieces (Round caps and joins, the closing bridge) are written directly.
template<class VS, class W>
VGX_EL void elem_emit(const MeshCtxT<VS>& m, const Elem& e, uint32_t b, uint32_t k, Rails prev, W& w)
{
	const uint32_t N = m.N;
	const uint32_t color = w.color, c0 = w.c0;
	const V2 p1 = e.p1;
	const float hsw = m.hsw, hswAA = m.hswAA, fringe = m.fringe;

	// ------------------------------- AA stroke, 4 rails -------------------------------------------
	if (m.kind == VGX_MESH_STROKE_AA) {
		if (e.et != ET_JOIN) {
			const bool firstCap = e.et == ET_CAP_FIRST;
			const V2 d = e.d01;
			const V2 l = v2ccw(d);
			if (m.cap == VGX_CAP_ROUND) { // :1475-1515, 1917-1968
				const uint32_t H = e.H;
				const float startAngle = vgm_atan2(l.y, l.x);
				for (uint32_t i = 0; i < H; ++i) {
					const float t = i * VGM_PI / (float)(H - 1);
					const float a = firstCap ? startAngle + t : startAngle - t;
					float sa, ca;
					vgm_sincos(a, &sa, &ca);
					w.v2(b + 2 * i, v2(p1.x + ca * hsw, p1.y + sa * hsw), color, v2(p1.x + ca * hswAA, p1.y + sa * hswAA), c0);
				}
				if (firstCap) {
					uint32_t q = k;
					for (uint32_t i = 0; i + 2 < H; ++i, q += 3) { w.tri(q, 0, (i << 1) + 2, (i << 1) + 4); }
					for (uint32_t i = 0; i + 1 < H; ++i, q += 6) {
						const uint32_t base = i << 1;
						w.tri(q, base, base + 1, base + 3);
						w.tri(q + 3, base, base + 3, base + 2);
					}
				} else {
					const uint32_t en = b + (H - 1) * 2;
					w.bridge4(k, prev, rails(b + 1, b, en, en + 1));
					uint32_t q = k + 18;
					for (uint32_t i = 0; i + 2 < H; ++i, q += 3) {
						const uint32_t base = b + (i << 1);
						w.tri(q, b, base + 4, base + 2);
					}
					for (uint32_t i = 0; i + 1 < H; ++i, q += 6) {
						const uint32_t base = b + (i << 1);
						w.tri(q, base, base + 3, base + 1);
						w.tri(q + 3, base, base + 2, base + 3);
					}
				}
				return;
			}
			const V2 lh = v2mul(l, hsw);
			const V2 lhaa = v2mul(l, hswAA);
			if (m.cap == VGX_CAP_BUTT) { // :1422-1447, 1858-1886
				const V2 daa = v2mul(d, fringe);
				if (firstCap) {
					w.sv(0, v2add(p1, v2sub(lhaa, daa)), c0);
					w.sv(1, v2add(p1, lh), color);
					w.sv(2, v2sub(p1, lh), color);
					w.sv(3, v2sub(p1, v2add(lhaa, daa)), c0);
				} else {
					w.sv(0, v2add(p1, v2add(lhaa, daa)), c0);
					w.sv(1, v2add(p1, lh), color);
					w.sv(2, v2sub(p1, lh), color);
					w.sv(3, v2sub(p1, v2sub(lhaa, daa)), c0);
				}
			} else { // Square, :1448-1474, 1887-1916
				const V2 dh = v2mul(d, hsw);
				const V2 dhaa = v2mul(d, hswAA);
				if (firstCap) {
					w.sv(0, v2add(p1, v2sub(lhaa, dhaa)), c0);
					w.sv(1, v2add(p1, v2sub(lh, dh)), color);
					w.sv(2, v2sub(p1, v2add(lh, dh)), color);
					w.sv(3, v2sub(p1, v2add(lhaa, dhaa)), c0);
				} else {
					w.sv(0, v2add(p1, v2add(lhaa, dhaa)), c0);
					w.sv(1, v2add(p1, v2add(lh, dh)), color);
					w.sv(2, v2sub(p1, v2sub(lh, dh)), color);
					w.sv(3, v2sub(p1, v2sub(lhaa, dhaa)), c0);
				}
			}
			if (firstCap) {
				w.stri(0, 0, 2, 1);
				w.stri(3, 0, 3, 2);
			} else {
				w.sbridge4(0, prev, rails(b, b + 1, b + 2, b + 3));
				w.stri(18, b, b + 1, b + 2);
				w.stri(21, b, b + 2, b + 3);
			}
			return;
		}
		// join, :1524-1850
		const V2 vhaa = v2mul(e.v, hswAA);
		const V2 vh = v2mul(e.v, hsw);
		const bool L = e.leftInner;
		const V2 innerAA = L ? v2add(p1, vhaa) : v2sub(p1, vhaa);
		const V2 inner = L ? v2add(p1, vh) : v2sub(p1, vh);
		const Rails entry = L ? rails(b, b + 1, b + 2, b + 3) : rails(b + 3, b + 2, b + 1, b);
		uint32_t q = k;
		w.sv(0, innerAA, c0);
		w.sv(1, inner, color);
		if (e.hasConnect) { w.sbridge4(0, prev, entry); q += 18; }
		if (m.join == VGX_JOIN_MITER) {
			w.sv(2, L ? v2sub(p1, vh) : v2add(p1, vh), color);
			w.sv(3, L ? v2sub(p1, vhaa) : v2add(p1, vhaa), c0);
		} else {
			const V2 n01 = L ? v2cw(e.d01) : v2ccw(e.d01);
			const V2 n12 = L ? v2cw(e.d12) : v2ccw(e.d12);
			const uint32_t n = e.arc.n;
			{
				V2 a = v2add(p1, v2mul(n01, hsw));
				const V2 aAA = v2add(p1, v2mul(n01, hswAA));
				if (m.join == VGX_JOIN_BEVEL) {
					const float cosAngle = vgm_abs(v2dot(n01, n12));
					a = v2sub(a, v2mul(e.d01, cosAngle * fringe));
				}
				w.sv(2, a, color);
				w.sv(3, aAA, c0);
			}
			for (uint32_t i = 1; i < n; ++i) {
				const float ang = e.arc.a01 + i * e.arc.arcDa;
				float sa, ca;
				vgm_sincos(ang, &sa, &ca);
				const V2 dir = v2(ca, sa);
				w.v2(b + 2 + 2 * i, v2add(p1, v2mul(dir, hsw)), color, v2add(p1, v2mul(dir, hswAA)), c0);
			}
			{
				V2 a = v2add(p1, v2mul(n12, hsw));
				const V2 aAA = v2add(p1, v2mul(n12, hswAA));
				if (m.join == VGX_JOIN_BEVEL) {
					const float cosAngle = vgm_abs(v2dot(n01, n12));
					a = v2add(a, v2mul(e.d12, cosAngle * fringe));
				}
				w.v2(b + 2 + 2 * n, a, color, aAA, c0);
			}
			uint32_t arcID = b + 2;
			for (uint32_t i = 0; i < n; ++i, arcID += 2, q += 9) {
				if (L) {
					w.tri3(q, b + 1, arcID, arcID + 2, arcID, arcID + 1, arcID + 3, arcID, arcID + 3, arcID + 2);
				} else {
					w.tri3(q, b + 1, arcID + 2, arcID, arcID, arcID + 3, arcID + 1, arcID, arcID + 2, arcID + 3);
				}
			}
		}
		if (e.closesLoop) { // :1970-1984
			const VgxJoin j0 = vgx_join(p1, m.vtx.ld(0), m.vtx.ld(N > 1 ? 1 : 0), hswAA);
			w.bridge4(q, elem_exit_rails(m, e, b), first_join_entry(m, j0.leftInner));
		}
		return;
	}

	// ------------------------------- non-AA stroke, 2 rails ---------------------------------------
	if (m.kind == VGX_MESH_STROKE) {
		if (e.et != ET_JOIN) {
			const bool firstCap = e.et == ET_CAP_FIRST;
			const V2 d = e.d01;
			const V2 l = v2ccw(d);
			if (m.cap == VGX_CAP_ROUND) { // :1059-1080, 1342-1370
				const uint32_t H = e.H;
				const float startAngle = vgm_atan2(l.y, l.x);
				for (uint32_t i = 0; i < H; ++i) {
					const float t = i * VGM_PI / (float)(H - 1);
					const float a = firstCap ? startAngle + t : startAngle - t;
					float sa, ca;
					vgm_sincos(a, &sa, &ca);
					w.v(b + i, v2(p1.x + ca * hsw, p1.y + sa * hsw), color);
				}
				if (firstCap) {
					uint32_t q = k;
					for (uint32_t i = 0; i + 2 < H; ++i, q += 3) { w.tri(q, 0, i + 1, i + 2); }
				} else {
					w.bridge2(k, prev, rails(b, b + (H - 1), 0, 0));
					uint32_t q = k + 6;
					for (uint32_t i = 0; i + 2 < H; ++i, q += 3) { w.tri(q, b, b + i + 2, b + i + 1); }
				}
				return;
			}
			const V2 lh = v2mul(l, hsw);
			if (m.cap == VGX_CAP_BUTT) { // :1032-1044, 1303-1321
				w.sv(0, v2add(p1, lh), color);
				w.sv(1, v2sub(p1, lh), color);
			} else { // Square :1045-1058, 1322-1341
				const V2 dh = v2mul(d, hsw);
				if (firstCap) {
					w.sv(0, v2add(p1, v2sub(lh, dh)), color);
					w.sv(1, v2sub(p1, v2add(lh, dh)), color);
				} else {
					w.sv(0, v2add(p1, v2add(lh, dh)), color);
					w.sv(1, v2sub(p1, v2sub(lh, dh)), color);
				}
			}
			if (!firstCap) { w.sbridge2(0, prev, rails(b, b + 1, 0, 0)); }
			return;
		}
		// join, :1088-1296
		const V2 vh = v2mul(e.v, hsw);
		const bool L = e.leftInner;
		const V2 inner = L ? v2add(p1, vh) : v2sub(p1, vh);
		const Rails entry = L ? rails(b, b + 1, 0, 0) : rails(b + 1, b, 0, 0);
		uint32_t q = k;
		w.sv(0, inner, color);
		if (e.hasConnect) { w.sbridge2(0, prev, entry); q += 6; }
		if (m.join == VGX_JOIN_MITER) {
			w.sv(1, L ? v2sub(p1, vh) : v2add(p1, vh), color);
		} else {
			const V2 n01 = L ? v2cw(e.d01) : v2ccw(e.d01);
			const V2 n12 = L ? v2cw(e.d12) : v2ccw(e.d12);
			const uint32_t n = e.arc.n;
			w.sv(1, v2add(p1, v2mul(n01, hsw)), color);
			for (uint32_t i = 1; i < n; ++i) {
				const float ang = e.arc.a01 + i * e.arc.arcDa;
				float sa, ca;
				vgm_sincos(ang, &sa, &ca);
				w.v(b + 1 + i, v2(p1.x + hsw * ca, p1.y + hsw * sa), color);
			}
			w.v(b + 1 + n, v2add(p1, v2mul(n12, hsw)), color);
			for (uint32_t i = 0; i < n; ++i, q += 3) {
				const uint32_t base = b + i;
				if (L) { w.tri(q, b, base + 1, base + 2); } else { w.tri(q, b, base + 2, base + 1); }
			}
		}
		if (e.closesLoop) { // :1372-1380
			const VgxJoin j0 = vgx_join(p1, m.vtx.ld(0), m.vtx.ld(N > 1 ? 1 : 0), hsw);
			w.bridge2(q, elem_exit_rails(m, e, b), first_join_entry(m, j0.leftInner));
		}
		return;
	}

	// ------------------------------- thin AA stroke, 3 rails --------------------------------------
	{
		const float f = fringe; // stroker.cpp:1999
		if (e.et != ET_JOIN) { // :2012-2058, 2242-2294
			const bool firstCap = e.et == ET_CAP_FIRST;
			const V2 d = e.d01;
			const V2 l = v2ccw(d);
			const V2 lf = v2mul(l, f);
			if (m.cap == VGX_CAP_BUTT) {
				w.sv(0, v2add(p1, lf), c0);
				w.sv(1, p1, color);
				w.sv(2, v2sub(p1, lf), c0);
			} else {
				const V2 df = v2mul(d, f);
				if (firstCap) {
					w.sv(0, v2add(p1, v2sub(lf, df)), c0);
					w.sv(1, p1, color);
					w.sv(2, v2sub(p1, v2add(lf, df)), c0);
				} else {
					w.sv(0, v2add(p1, v2add(lf, df)), c0);
					w.sv(1, p1, color);
					w.sv(2, v2sub(p1, v2sub(lf, df)), c0);
				}
			}
			if (!firstCap) { w.sbridge3(0, prev, rails(b, b + 1, b + 2, 0)); }
			return;
		}
		const V2 vf = v2mul(e.v, f);
		const bool L = e.leftInner;
		const V2 inner = L ? v2add(p1, vf) : v2sub(p1, vf);
		const Rails entry = L ? rails(b, b + 1, b + 2, 0) : rails(b + 2, b + 1, b, 0);
		uint32_t q = k;
		w.sv(0, inner, c0);
		w.sv(1, p1, color);
		if (e.hasConnect) { w.sbridge3(0, prev, entry); q += 12; }
		if (m.join == VGX_JOIN_MITER) {
			w.sv(2, L ? v2sub(p1, vf) : v2add(p1, vf), c0);
		} else {
			const V2 n01 = L ? v2cw(e.d01) : v2ccw(e.d01);
			const V2 n12 = L ? v2cw(e.d12) : v2ccw(e.d12);
			w.sv(2, v2add(p1, v2mul(n01, f)), c0);
			w.sv(3, v2add(p1, v2mul(n12, f)), c0);
			if (L) { w.tri(q, b + 1, b + 2, b + 3); } else { w.tri(q, b + 1, b + 3, b + 2); }
			q += 3;
		}
		if (e.closesLoop) { // :2295-2306
			const VgxJoin j0 = vgx_join(p1, m.vtx.ld(0), m.vtx.ld(N > 1 ? 1 : 0), f);
			w.bridge3(q, elem_exit_rails(m, e, b), first_join_entry(m, j0.leftInner));
		}
	}
}

// ---- per-mesh preparation ------------------------------------------------------------------------------
VGX_EL VgxMeshPrep mesh_prep(const VgxMeshDesc& md, const vgx_draw* dr, const float* poly)
{
	VgxMeshPrep pr;
	const uint32_t kind = VGX_MD_KIND(md.kind);
	if (kind >= VGX_MESH_STROKE) {
		const VgxStrokeParams sp = vgx_stroke_params(kind, VGX_MD_CLOSED(md.kind) != 0, dr->stroke_flags, dr->stroke_width, dr->fringe, dr->scale, dr->tess_tol);
		pr.f0 = sp.hsw; pr.f1 = sp.hswAA; pr.f2 = dr->fringe;
		pr.color = dr->stroke_color;
	} else {
		pr.f0 = 0.0f; pr.f1 = 0.0f; pr.f2 = dr->fringe;
		if (kind == VGX_MESH_FILL_AA) {
			// orientation from the first triangle only (stroker.cpp:721-723, "might not work in all cases")
			const float* vtx = poly + 2 * md.poly_first;
			const V2 q0 = ldv(vtx, 0), q1 = ldv(vtx, 1), q2 = ldv(vtx, 2);
			const float orient = v2cross(v2sub(q1, q0), v2sub(q2, q0));
			pr.f0 = dr->fringe * 0.5f * vgm_sign(orient);
		}
		pr.color = dr->fill_color;
	}
	return pr;
}

// ------------------------------------------------------------------------------------------------
// Polyline strokes: one 64-element chunk of strokerPolylineStroke / StrokeAA / StrokeAAThin (stroker.cpp:1008-2314).
// The caller found every lane's mesh (mc: constants, element index j, vertex source) and output pointers; this does
//   A. geometry + own vertex / index counts (one vertex read and one vec2Dir per element; neighbours come from the
//      adjacent lanes), B. bases inside the mesh (wave prefix scan segmented by mesh, carried across chunks),
//   C. previous element's exit rails (prevSegment*ID, stroker.cpp:1401-1410), D. the stores.
// laneHasNext: the next lane holds the next element of the same contiguous element range.
// ------------------------------------------------------------------------------------------------
struct StrokeCarry
{
	uint32_t v, i; uint64_t rails;
#ifdef VGX_STROKE_PROFILE
	unsigned long long tw, tg, te; // clocks: the wait for a chunk's vertices (behind the stores of the chunk in front), geometry + scans, emit + copy-out
#endif
};

// LDS stage of one chunk's output (k_stroke, round 6): a chunk whose elements all belong to ONE mesh writes its vertices / colours /
// indices into LDS (the element code is the same: its stores go through generic pointers that point there) and the wave copies them
// out as dense runs. Why: with per-lane stores a Round-join chunk is ~15 write requests per element of ~11 bytes each (PMC on BASELINE
// configs[3]: 137 M TCP -> TCC write requests for 1.9 GB, the L1 stalled 80 % of the kernel's time, profiles/r06_pmc_sq_round10k.txt);
// copied out 16 bytes per lane, consecutive lanes consecutive, the same bytes are ~1/4 of the requests.
// SCOL / SIDX: capacities of the stage in vertices / indices (0: the stream is stored directly). k_stroke_long (vgx_stroke.hip) stages
// colours and indices -- 11 of a Round-join element's 15 requests -- in 8.9 KB; positions too would leave two waves per SIMD
// (measured, same box, BASELINE configs[3]: nothing staged 0.735 ms, indices 0.617, indices + colours 0.585, all three 0.665).
template<int SCOL, int SIDX>
struct __attribute__((aligned(16))) StrokeStageT
{
	uint32_t col[SCOL ? SCOL : 4];
	uint16_t idx[SIDX ? SIDX : 8];
};

template<class VS, int SCOL = 0, int SIDX = 0>
__device__ __forceinline__ void stroke_chunk(bool valid, bool laneHasNext, int nvalid, int lane, const MeshCtxT<VS>& mc, uint32_t color,
	float* posMesh, uint32_t* colMesh, uint16_t* idxMesh, uint32_t idxBase, StrokeCarry& carry, StrokeStageT<SCOL, SIDX>* stage = nullptr, bool oneMesh = false)
{
	// step A
#ifdef VGX_STROKE_PROFILE
	const unsigned long long tp0 = clock64();
#endif
	V2 p1 = v2(0.0f, 0.0f);
	if (valid) { p1 = mc.vtx.ld(mc.j); }
	const bool prevInWave = lane > 0 && mc.j > 0;
	const bool nextInWave = laneHasNext && mc.j + 1 < mc.N;
#ifdef VGX_STROKE_PROFILE
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (waits for the stores of the chunk in front as well: one in-order counter)
	const unsigned long long tp1 = clock64();
#endif
	V2 pNext;
	pNext.x = wave_from_next(p1.x, 0.0f); pNext.y = wave_from_next(p1.y, 0.0f);
	if (valid && !nextInWave) { pNext = mc.vtx.ld(mc.j + 1 < mc.N ? mc.j + 1 : 0); }
	V2 d12 = v2(0.0f, 0.0f);
	if (valid) { d12 = v2dir(p1, pNext); }
	V2 dPrev;
	dPrev.x = wave_from_prev(d12.x, 0.0f); dPrev.y = wave_from_prev(d12.y, 0.0f);
	if (valid && !prevInWave) { dPrev = v2dir(mc.vtx.ld(mc.j > 0 ? mc.j - 1 : mc.N - 1), p1); }
	Elem e;
	e.nv = 0; e.ni = 0; e.et = ET_JOIN; e.leftInner = true; e.hasConnect = false; e.closesLoop = false;
	e.arc.a01 = 0.0f; e.arc.arcDa = 0.0f; e.arc.n = 1; e.H = 2; e.p1 = p1; e.d01 = dPrev; e.d12 = d12; e.v = d12;
	uint32_t totalIdx = 0;
	if (valid) {
		e = elem_geometry(mc, p1, dPrev, d12);
		totalIdx = elem_total_indices(mc, e);
	}

	// step B: running vertex / index counters of the mesh (segmented wave scan)
	const uint64_t heads = wave_ballot(valid && mc.j == 0);
	const int mh = seg_head(heads, lane);
	const uint32_t inclV = wave_incl_scan_u32(e.nv, lane);
	const uint32_t inclI = wave_incl_scan_u32(totalIdx, lane);
	const uint32_t exV = inclV - e.nv, exI = inclI - totalIdx;
	const uint32_t hV = wave_read_u32(exV, mh < 0 ? 0 : mh), hI = wave_read_u32(exI, mh < 0 ? 0 : mh);
	const uint32_t vbase = mh < 0 ? carry.v + exV : exV - hV;
	const uint32_t ibase = mh < 0 ? carry.i + exI : exI - hI;

	// step C: previous element's exit rails
	const uint64_t myExit = valid ? rails_pack(elem_exit_rails(mc, e, vbase)) : 0ull;
	const uint64_t prevPacked = (uint64_t)wave_from_prev_u32((uint32_t)myExit, (uint32_t)carry.rails) | ((uint64_t)wave_from_prev_u32((uint32_t)(myExit >> 32), (uint32_t)(carry.rails >> 32)) << 32);

	// step D
	const bool meshLast = valid && (mc.j == mc.N - 1);
	const int Lz = nvalid - 1;
	const uint32_t endV = wave_read_u32(vbase + e.nv, Lz);
	const uint32_t endI = wave_read_u32(ibase + totalIdx, Lz);
	// the chunk's output ranges inside its mesh (one mesh: lane 0 holds the first element): [v0, endV) vertices, [i0, endI) indices
	const uint32_t v0 = wave_read_u32(vbase, 0), i0 = wave_read_u32(ibase, 0);
	const bool staged = (SCOL || SIDX) && stage != nullptr && oneMesh && nvalid == VGX_WAVE
		&& (!SCOL || endV - v0 <= (uint32_t)SCOL) && (!SIDX || endI - i0 <= (uint32_t)SIDX); // wave-uniform
#ifdef VGX_STROKE_PROFILE
	const unsigned long long tp2 = clock64();
#endif
	if (valid) {
		StrokeWriter w;
		w.pos = posMesh;
		w.col = colMesh;
		w.idx = idxMesh;
		if (staged) { // generic pointers at the stage + the chunk's first vertex / index as the writer's position bias (a biased POINTER would not do: the
			// compiler knows the stage is LDS and does that arithmetic in 32 bits)
			w.vo = v0; w.io = i0;
			w.pos = posMesh + 2 * (size_t)v0;
			w.col = SCOL ? (uint32_t*)stage->col : colMesh + (size_t)v0;
			w.idx = SIDX ? (uint16_t*)stage->idx : idxMesh + (size_t)i0;
		}
		w.color = color;
		w.c0 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
		w.ib = idxBase;
		w.reset();
		elem_emit(mc, e, vbase, ibase, rails_unpack(prevPacked), w);
		w.flush(vbase, ibase);
	}
	if (staged) { // wave-uniform: dense copy-out, 16 bytes per lane and step
		__syncthreads(); // (one-wave workgroup: the LDS writes above have landed)
		const uint64_t cm = wave_bcast_u64((uint64_t)colMesh, 0), im = wave_bcast_u64((uint64_t)idxMesh, 0);
		if (SCOL) {
			const uint32_t bytes = (endV - v0) * 4u;
			char* dst = (char*)((uint32_t*)cm + (size_t)v0);
			const char* src = (const char*)stage->col;
			for (uint32_t o = (uint32_t)lane * 16u; o < bytes; o += VGX_WAVE * 16u) {
				if (o + 16u <= bytes) { struct __attribute__((packed, aligned(4))) C4 { uint32_t a, b, c, d; }; *(C4*)(dst + o) = *(const C4*)(src + o); }
				else { for (uint32_t q = o; q < bytes; q += 4u) { *(uint32_t*)(dst + q) = *(const uint32_t*)(src + q); } }
			}
		}
		if (SIDX) {
			const uint32_t bytes = (endI - i0) * 2u;
			char* dst = (char*)((uint16_t*)im + (size_t)i0);
			const char* src = (const char*)stage->idx;
			for (uint32_t o = (uint32_t)lane * 16u; o < bytes; o += VGX_WAVE * 16u) {
				if (o + 16u <= bytes) { struct __attribute__((packed, aligned(2))) I8 { uint32_t a, b, c, d; }; *(I8*)(dst + o) = *(const I8*)(src + o); }
				else { for (uint32_t q = o; q < bytes; q += 2u) { *(uint16_t*)(dst + q) = *(const uint16_t*)(src + q); } }
			}
		}
		__syncthreads(); // the stage is free for the next chunk
	}

#ifdef VGX_STROKE_PROFILE
	carry.tw += tp1 - tp0; carry.tg += tp2 - tp1; carry.te += clock64() - tp2;
#endif
	// carries (from the last valid lane)
	const int lastIsMeshLast = wave_bcast((int)meshLast, Lz);
	const uint64_t endRails = wave_bcast_u64(myExit, Lz);
	carry.v = lastIsMeshLast ? 0u : endV;
	carry.i = lastIsMeshLast ? 0u : endI;
	carry.rails = lastIsMeshLast ? 0ull : endRails;
}

// The same chunk when every element in it belongs to a CLOSED stroke with MITER joins, AA (4 rails) or Thin (3 rails) --
// e.g. every stroke of the tiger: each element is one join with R vertices at R j and, from the second element on, one
// bridge of 6 (R - 1) indices at 6 (R - 1) (j - 1); the last element adds the closing bridge (stroker.cpp:1524-1579,
// 1970-1984; thin :2060-2110, 2295-2306). Nothing is data dependent, so steps B (segmented scans of the counts) and the
// register stage of step D fall away: positions, colours and indices are computed and stored directly, same values and
// same addresses as stroke_chunk (which handles a chunk as soon as one of its elements is anything else; the carry
// both maintain makes the two interchangeable chunk by chunk). 568 -> ~250 VALU instructions per chunk on the tiger.
VGX_EL bool stroke_elem_is_simple(uint32_t kind, bool closed, uint32_t join)
{
	return closed && join == VGX_JOIN_MITER && (kind == VGX_MESH_STROKE_AA || kind == VGX_MESH_STROKE_AA_THIN);
}

template<class VS>
__device__ __forceinline__ void stroke_chunk_simple(bool valid, bool laneHasNext, int nvalid, int lane, const MeshCtxT<VS>& mc, uint32_t color,
	float* posMesh, uint32_t* colMesh, uint16_t* idxMesh, uint32_t idxBase, StrokeCarry& carry)
{
	// step A, as in stroke_chunk
	V2 p1 = v2(0.0f, 0.0f);
	if (valid) { p1 = mc.vtx.ld(mc.j); }
	const bool prevInWave = lane > 0 && mc.j > 0;
	const bool nextInWave = laneHasNext && mc.j + 1 < mc.N;
	V2 pNext;
	pNext.x = wave_from_next(p1.x, 0.0f); pNext.y = wave_from_next(p1.y, 0.0f);
	if (valid && !nextInWave) { pNext = mc.vtx.ld(mc.j + 1 < mc.N ? mc.j + 1 : 0); }
	V2 d12 = v2(0.0f, 0.0f);
	if (valid) { d12 = v2dir(p1, pNext); }
	V2 dPrev;
	dPrev.x = wave_from_prev(d12.x, 0.0f); dPrev.y = wave_from_prev(d12.y, 0.0f);
	if (valid && !prevInWave) { dPrev = v2dir(mc.vtx.ld(mc.j > 0 ? mc.j - 1 : mc.N - 1), p1); }

	const bool thin = mc.kind == VGX_MESH_STROKE_AA_THIN;
	const uint32_t R = thin ? 3u : 4u;
	const uint32_t bridgeIdx = thin ? 12u : 18u;
	const float sideWidth = thin ? mc.fringe : mc.hswAA; // elem_geometry
	const VgxJoin jn = vgx_join_dirs(dPrev, d12, sideWidth);
	const bool L = jn.leftInner;
	const uint32_t N = mc.N, j = mc.j;
	const uint32_t b = R * j;
	// entry = exit rails of a Miter join (elem_exit_rails)
	const uint32_t top = b + R - 1;
	const Rails mine = thin ? (L ? rails(b, b + 1, b + 2, 0) : rails(top, b + 1, b, 0)) : (L ? rails(b, b + 1, b + 2, b + 3) : rails(top, b + 2, b + 1, b));
	const uint64_t myExit = valid ? rails_pack(mine) : 0ull;
	const uint64_t prevPacked = (uint64_t)wave_from_prev_u32((uint32_t)myExit, (uint32_t)carry.rails) | ((uint64_t)wave_from_prev_u32((uint32_t)(myExit >> 32), (uint32_t)(carry.rails >> 32)) << 32);

	const bool meshLast = valid && (j == N - 1);
	if (valid) {
		const uint32_t c0 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
		float* pp = posMesh + 2 * (size_t)b;
		uint32_t* pc = colMesh + b;
		if (thin) { // stroker.cpp:2060-2110
			const V2 vf = v2mul(jn.v, mc.fringe);
			const V2 q0 = L ? v2add(p1, vf) : v2sub(p1, vf);
			const V2 q2 = L ? v2sub(p1, vf) : v2add(p1, vf);
			PosPair q; q.x0 = q0.x; q.y0 = q0.y; q.x1 = p1.x; q.y1 = p1.y;
			*(PosPair*)pp = q;
			*(float2*)(pp + 4) = make_float2(q2.x, q2.y);
			ColPair c; c.c0 = c0; c.c1 = color;
			*(ColPair*)pc = c;
			pc[2] = c0;
		} else { // :1524-1579
			const V2 vhaa = v2mul(jn.v, mc.hswAA);
			const V2 vh = v2mul(jn.v, mc.hsw);
			const V2 q0 = L ? v2add(p1, vhaa) : v2sub(p1, vhaa);
			const V2 q1 = L ? v2add(p1, vh) : v2sub(p1, vh);
			const V2 q2 = L ? v2sub(p1, vh) : v2add(p1, vh);
			const V2 q3 = L ? v2sub(p1, vhaa) : v2add(p1, vhaa);
			PosPair q; q.x0 = q0.x; q.y0 = q0.y; q.x1 = q1.x; q.y1 = q1.y;
			*(PosPair*)pp = q;
			PosPair r; r.x0 = q2.x; r.y0 = q2.y; r.x1 = q3.x; r.y1 = q3.y;
			*(PosPair*)(pp + 4) = r;
			ColPair c; c.c0 = c0; c.c1 = color;
			*(ColPair*)pc = c;
			ColPair d; d.c0 = color; d.c1 = c0;
			*(ColPair*)(pc + 2) = d;
		}
		const uint32_t ib = idxBase;
		if (j > 0) { // the bridge from the previous join (bridge4 / bridge3 of the writer)
			const Rails p = rails_unpack(prevPacked);
			const uint32_t pa = p.a + ib, pb = p.b + ib, pcc = p.c + ib, pd = p.d + ib;
			const uint32_t ca = mine.a + ib, cb = mine.b + ib, cc = mine.c + ib, cd = mine.d + ib;
			uint16_t* pi = idxMesh + (size_t)bridgeIdx * (j - 1);
			Idx6 t0; t0.a = (pa & 0xFFFFu) | (pb << 16); t0.b = (cb & 0xFFFFu) | (pa << 16); t0.c = (cb & 0xFFFFu) | (ca << 16);
			Idx6 t1; t1.a = (pb & 0xFFFFu) | (pcc << 16); t1.b = (cc & 0xFFFFu) | (pb << 16); t1.c = (cc & 0xFFFFu) | (cb << 16);
			*(Idx6*)pi = t0;
			*(Idx6*)(pi + 6) = t1;
			if (!thin) {
				Idx6 t2; t2.a = (pcc & 0xFFFFu) | (pd << 16); t2.b = (cd & 0xFFFFu) | (pcc << 16); t2.c = (cd & 0xFFFFu) | (cc << 16);
				*(Idx6*)(pi + 12) = t2;
			}
		}
		if (meshLast) { // closing bridge to join 0 (:1970-1984, 2295-2306); d12 = vec2Dir(last vertex, vertex 0) already
			const V2 v0 = pNext;
			const V2 v1 = mc.vtx.ld(N > 1 ? 1 : 0);
			const VgxJoin j0 = vgx_join_dirs(d12, v2dir(v0, v1), sideWidth);
			const Rails f = thin ? (j0.leftInner ? rails(0, 1, 2, 0) : rails(2, 1, 0, 0)) : (j0.leftInner ? rails(0, 1, 2, 3) : rails(3, 2, 1, 0));
			const uint32_t pa = mine.a + ib, pb = mine.b + ib, pcc = mine.c + ib, pd = mine.d + ib;
			const uint32_t ca = f.a + ib, cb = f.b + ib, cc = f.c + ib, cd = f.d + ib;
			uint16_t* pi = idxMesh + (size_t)bridgeIdx * (N - 1);
			Idx6 t0; t0.a = (pa & 0xFFFFu) | (pb << 16); t0.b = (cb & 0xFFFFu) | (pa << 16); t0.c = (cb & 0xFFFFu) | (ca << 16);
			Idx6 t1; t1.a = (pb & 0xFFFFu) | (pcc << 16); t1.b = (cc & 0xFFFFu) | (pb << 16); t1.c = (cc & 0xFFFFu) | (cb << 16);
			*(Idx6*)pi = t0;
			*(Idx6*)(pi + 6) = t1;
			if (!thin) {
				Idx6 t2; t2.a = (pcc & 0xFFFFu) | (pd << 16); t2.b = (cd & 0xFFFFu) | (pcc << 16); t2.c = (cd & 0xFFFFu) | (cc << 16);
				*(Idx6*)(pi + 12) = t2;
			}
		}
	}

	// carries, in stroke_chunk's terms: vertices / indices of the mesh written so far, exit rails of the last element
	const int Lz = nvalid - 1;
	const int lastIsMeshLast = wave_bcast((int)meshLast, Lz);
	const uint32_t endV = wave_read_u32(b + R, Lz);
	const uint32_t endI = wave_read_u32(bridgeIdx * j, Lz);
	const uint64_t endRails = wave_bcast_u64(myExit, Lz);
	carry.v = lastIsMeshLast ? 0u : endV;
	carry.i = lastIsMeshLast ? 0u : endI;
	carry.rails = lastIsMeshLast ? 0ull : endRails;
}

// Vertex / index counts of a whole stroke mesh with Round joins (the only data-dependent sizes, numArcPoints per join,
// stroker.cpp:1146, 1592): the wave's lanes stride over the mesh's elements and sum what stroke_chunk will emit for
// each. 64-bit sums, saturated: a hostile draw (arc counts saturate at 131072 per join) cannot wrap back into range.
template<class VS>
__device__ __forceinline__ void round_mesh_size(MeshCtxT<VS> mc, int lane, uint32_t* nvOut, uint32_t* niOut)
{
	const uint32_t N = mc.N;
	uint64_t nv = 0, ni = 0;
	mc.da = mesh_da(mc); // once per mesh, not once per element (an acos and two draw-record loads each)
	for (uint32_t j = lane; j < N; j += VGX_WAVE) {
		mc.j = j;
		const V2 p1 = mc.vtx.ld(j);
		const V2 d12 = v2dir(p1, mc.vtx.ld(j + 1 < N ? j + 1 : 0));
		const V2 dPrev = v2dir(mc.vtx.ld(j > 0 ? j - 1 : N - 1), p1);
		const Elem e = elem_geometry(mc, p1, dPrev, d12);
		nv += e.nv;
		ni += elem_total_indices(mc, e);
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		nv += __shfl_xor((unsigned long long)nv, d);
		ni += __shfl_xor((unsigned long long)ni, d);
	}
	*nvOut = vgx_sat_nv(nv);
	*niOut = vgx_sat_ni(ni);
}

// ------------------------------------------------------------------------------------------------
// AA strokes with Round joins (strokerPolylineStrokeAA with LineJoin::Round, stroker.cpp:1580-1691; caps :1419-1515, 1856-1968): the
// element code of elem_geometry + elem_emit cut down to this one style -- own join (inner pair, the arc's first pair, one sincos pair per
// inner arc point, the arc's last pair; nine indices per arc segment) and the bridge from the previous element's exit rails --, with
// direct wide stores. Used by the template kernels (vgx_tmpl.hip: tmpl_stroke_elem_round); as a chunk routine of k_stroke it ran no faster than the
// general one (profiles/experiments/r05_k_stroke_round_aa.patch). pp / pc: the element's first vertex in the position / colour stream, bi: its number as an index VALUE
// (mesh-relative + the assembly base), prev: the previous element's exit rails as index values.
// ------------------------------------------------------------------------------------------------
VGX_EL void raa_bridge(char* at, Rails p, Rails c) // bridge4 of the writer
{
	Idx6 t0; t0.a = (p.a & 0xFFFFu) | (p.b << 16); t0.b = (c.b & 0xFFFFu) | (p.a << 16); t0.c = (c.b & 0xFFFFu) | (c.a << 16);
	Idx6 t1; t1.a = (p.b & 0xFFFFu) | (p.c << 16); t1.b = (c.c & 0xFFFFu) | (p.b << 16); t1.c = (c.c & 0xFFFFu) | (c.b << 16);
	Idx6 t2; t2.a = (p.c & 0xFFFFu) | (p.d << 16); t2.b = (c.d & 0xFFFFu) | (p.c << 16); t2.c = (c.d & 0xFFFFu) | (c.c << 16);
	VGX_ST_GUARD(t0.a ^ t1.c) { *(Idx6*)at = t0; *(Idx6*)(at + 12) = t1; *(Idx6*)(at + 24) = t2; }
}
VGX_EL void raa_tri(char* at, uint32_t a0, uint32_t a1, uint32_t a2)
{
	Idx3 t; t.a = (a0 & 0xFFFFu) | (a1 << 16); t.b = (uint16_t)a2;
	VGX_ST_GUARD(t.a) { *(Idx3*)at = t; }
}
VGX_EL Rails raa_join_entry(uint32_t bi, bool L) { return L ? rails(bi, bi + 1, bi + 2, bi + 3) : rails(bi + 3, bi + 2, bi + 1, bi); }
VGX_EL Rails raa_join_exit(uint32_t bi, uint32_t n, bool L) { const uint32_t pe = bi + 2u + 2u * n; return L ? rails(bi, bi + 1, pe, pe + 1) : rails(pe + 1, pe, bi + 1, bi); }
VGX_EL Rails raa_cap_first_exit(uint32_t bi, uint32_t cap, uint32_t H) { return cap == VGX_CAP_ROUND ? rails(bi + 1, bi, bi + (H - 1) * 2, bi + (H - 1) * 2 + 1) : rails(bi, bi + 1, bi + 2, bi + 3); }
// one join: pown = where its own triangles go, pbridge = where the bridge prev -> entry goes (nullptr: none)
VGX_EL void raa_join_emit(char* pp, char* pc, char* pown, char* pbridge, uint32_t bi, uint32_t color, float hsw, float hswAA, V2 p1, const VgxJoin& jn, const VgxArc& arc, uint32_t n, Rails prev)
{
	const bool L = jn.leftInner;
	const uint32_t c0 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
	const V2 n01 = L ? v2cw(jn.d01) : v2ccw(jn.d01);
	const V2 n12 = L ? v2cw(jn.d12) : v2ccw(jn.d12);
	const V2 vhaa = v2mul(jn.v, hswAA);
	const V2 vh = v2mul(jn.v, hsw);
	const V2 q0 = L ? v2add(p1, vhaa) : v2sub(p1, vhaa);
	const V2 q1 = L ? v2add(p1, vh) : v2sub(p1, vh);
	const V2 q2 = v2add(p1, v2mul(n01, hsw));
	const V2 q3 = v2add(p1, v2mul(n01, hswAA));
	const V2 qa = v2add(p1, v2mul(n12, hsw));
	const V2 qb = v2add(p1, v2mul(n12, hswAA));
	PosPair q; q.x0 = q0.x; q.y0 = q0.y; q.x1 = q1.x; q.y1 = q1.y;
	PosPair r; r.x0 = q2.x; r.y0 = q2.y; r.x1 = q3.x; r.y1 = q3.y;
	PosPair u; u.x0 = qa.x; u.y0 = qa.y; u.x1 = qb.x; u.y1 = qb.y;
	ColPair c; c.c0 = c0; c.c1 = color;
	ColPair d; d.c0 = color; d.c1 = c0;
	VGX_ST_GUARD(c0 ^ __float_as_uint(q.x0) ^ __float_as_uint(u.y1)) {
	*(PosPair*)pp = q;
	*(PosPair*)(pp + 16) = r;
	*(PosPair*)(pp + 16 + 16 * n) = u;
	*(ColPair*)pc = c;
	*(ColPair*)(pc + 8) = d;
	*(ColPair*)(pc + 8 + 8 * n) = d;
	}
	for (uint32_t i = 1; i < n; ++i) { // the arc's inner points (:1610-1627)
		const float ang = arc.a01 + i * arc.arcDa;
		float sa, ca;
#ifdef VGX_EXP_FAKESIN
		sa = ang; ca = 1.0f - ang;
#else
		vgm_sincos(ang, &sa, &ca);
#endif
		const V2 dir = v2(ca, sa);
		const V2 w0 = v2add(p1, v2mul(dir, hsw)), w1 = v2add(p1, v2mul(dir, hswAA));
		PosPair t; t.x0 = w0.x; t.y0 = w0.y; t.x1 = w1.x; t.y1 = w1.y;
		VGX_ST_GUARD(c0 ^ __float_as_uint(t.x0)) {
		*(PosPair*)(pp + 16 + 16 * i) = t;
		*(ColPair*)(pc + 8 + 8 * i) = d;
		}
	}
	uint32_t a = bi + 2u; // arcID
	for (uint32_t i = 0; i < n; ++i, a += 2u, pown += 18) {
		Idx9 t; // tri3 of the writer
		if (L) {
			t.a = ((bi + 1u) & 0xFFFFu) | (a << 16); t.b = ((a + 2u) & 0xFFFFu) | (a << 16);
			t.c = ((a + 1u) & 0xFFFFu) | ((a + 3u) << 16); t.d = (a & 0xFFFFu) | ((a + 3u) << 16);
			t.e = (uint16_t)(a + 2u);
		} else {
			t.a = ((bi + 1u) & 0xFFFFu) | ((a + 2u) << 16); t.b = (a & 0xFFFFu) | (a << 16);
			t.c = ((a + 3u) & 0xFFFFu) | ((a + 1u) << 16); t.d = (a & 0xFFFFu) | ((a + 2u) << 16);
			t.e = (uint16_t)(a + 3u);
		}
		VGX_ST_GUARD(t.a ^ t.d) { *(Idx9*)pown = t; }
	}
	if (pbridge) { raa_bridge(pbridge, prev, raa_join_entry(bi, L)); }
}
// one cap of an open stroke: the first cap's own triangles at pi; the last cap's bridge (prev -> its entry) at pi, its own triangles behind
VGX_EL void raa_cap_emit(char* pp, char* pc, char* pi, uint32_t cap, bool first, uint32_t bi, uint32_t color, float hsw, float hswAA, float fringe, V2 p1, V2 d, uint32_t H, Rails prev)
{
	const uint32_t c0 = color & 0x00FFFFFFu;
	const V2 l = v2ccw(d);
	ColPair cd; cd.c0 = color; cd.c1 = c0;
	if (cap == VGX_CAP_ROUND) {
		const float startAngle = vgm_atan2(l.y, l.x);
		for (uint32_t i = 0; i < H; ++i) {
			const float t = i * VGM_PI / (float)(H - 1);
			const float a = first ? startAngle + t : startAngle - t;
			float sa, ca;
			vgm_sincos(a, &sa, &ca);
			PosPair q; q.x0 = p1.x + ca * hsw; q.y0 = p1.y + sa * hsw; q.x1 = p1.x + ca * hswAA; q.y1 = p1.y + sa * hswAA;
			VGX_ST_GUARD(c0 ^ __float_as_uint(q.x0)) { *(PosPair*)(pp + 16 * i) = q; *(ColPair*)(pc + 8 * i) = cd; }
		}
		if (first) { // fan + fringe quads
			for (uint32_t i = 0; i + 2 < H; ++i, pi += 6) { raa_tri(pi, bi, bi + (i << 1) + 2, bi + (i << 1) + 4); }
			for (uint32_t i = 0; i + 1 < H; ++i, pi += 12) {
				const uint32_t base = bi + (i << 1);
				raa_tri(pi, base, base + 1, base + 3);
				raa_tri(pi + 6, base, base + 3, base + 2);
			}
		} else {
			const uint32_t en = bi + (H - 1) * 2;
			raa_bridge(pi, prev, rails(bi + 1, bi, en, en + 1));
			pi += 36;
			for (uint32_t i = 0; i + 2 < H; ++i, pi += 6) { const uint32_t base = bi + (i << 1); raa_tri(pi, bi, base + 4, base + 2); }
			for (uint32_t i = 0; i + 1 < H; ++i, pi += 12) {
				const uint32_t base = bi + (i << 1);
				raa_tri(pi, base, base + 3, base + 1);
				raa_tri(pi + 6, base, base + 2, base + 3);
			}
		}
		return;
	}
	const V2 lh = v2mul(l, hsw), lhaa = v2mul(l, hswAA);
	V2 v0, v1, v2_, v3;
	if (cap == VGX_CAP_BUTT) {
		const V2 daa = v2mul(d, fringe);
		v0 = first ? v2add(p1, v2sub(lhaa, daa)) : v2add(p1, v2add(lhaa, daa));
		v1 = v2add(p1, lh); v2_ = v2sub(p1, lh);
		v3 = first ? v2sub(p1, v2add(lhaa, daa)) : v2sub(p1, v2sub(lhaa, daa));
	} else { // Square
		const V2 dh = v2mul(d, hsw), dhaa = v2mul(d, hswAA);
		v0 = first ? v2add(p1, v2sub(lhaa, dhaa)) : v2add(p1, v2add(lhaa, dhaa));
		v1 = first ? v2add(p1, v2sub(lh, dh)) : v2add(p1, v2add(lh, dh));
		v2_ = first ? v2sub(p1, v2add(lh, dh)) : v2sub(p1, v2sub(lh, dh));
		v3 = first ? v2sub(p1, v2add(lhaa, dhaa)) : v2sub(p1, v2sub(lhaa, dhaa));
	}
	PosPair q; q.x0 = v0.x; q.y0 = v0.y; q.x1 = v1.x; q.y1 = v1.y;
	PosPair r; r.x0 = v2_.x; r.y0 = v2_.y; r.x1 = v3.x; r.y1 = v3.y;
	ColPair c; c.c0 = c0; c.c1 = color;
	VGX_ST_GUARD(c0 ^ __float_as_uint(q.x0) ^ __float_as_uint(r.y1)) {
	*(PosPair*)pp = q; *(PosPair*)(pp + 16) = r;
	*(ColPair*)pc = c; *(ColPair*)(pc + 8) = cd;
	}
	if (first) {
		raa_tri(pi, bi, bi + 2, bi + 1);
		raa_tri(pi + 6, bi, bi + 3, bi + 2);
	} else {
		raa_bridge(pi, prev, rails(bi, bi + 1, bi + 2, bi + 3));
		raa_tri(pi + 36, bi, bi + 1, bi + 2);
		raa_tri(pi + 42, bi, bi + 2, bi + 3);
	}
}

// ------------------------------------------------------------------------------------------------
// Convex fills: strokerConvexFill / strokerConvexFillAA (stroker.cpp:334-365, 713-807), one polygon corner per lane.
//   FILL_AA element j: vertices 2j (inner, colour c) and 2j+1 (outer, colour c0) and the 9 (last element: 3)
//   index positions [9j, 9j+9) of the mesh, whose values are closed-form in (N, position).
//   FILL element j: vertex j (= polyline vertex) and fan triangle (0, j+1, j+2).
// ------------------------------------------------------------------------------------------------
// Everything one fill element needs, fetched one chunk AHEAD of its use (k_fill is a dependent chain per chunk:
// owner search -> vertex load -> stores; with the next chunk's loads already in flight while the current chunk
// computes and stores, a wave keeps two vertex loads outstanding instead of one).
struct FillFetch
{
	bool valid, aaElem, nextInWave, prevInWave, sseOrder;
	uint32_t j, N, color, ibase;
	float aa;
	uint64_t firstV, firstI, mi;
	V2 p1, pNextB, pPrevB;
};

// The nine index values [9j, 9j+9) of a convex AA fill mesh with N corners (element j), scalar or SSE order.
VGX_EL void fill_idx9(uint32_t j, uint32_t N, uint32_t ibase, bool sseOrder, uint32_t* val)
{
	struct { uint32_t ibase; bool sseOrder; } F; F.ibase = ibase; F.sseOrder = sseOrder;
#ifdef VGX_EXP_CHEAPMATH
			for (uint32_t g = 0; g < 9; ++g) { val[g] = (j + g) & 0xFFFFu; }
#else
#pragma unroll
			for (uint32_t g = 0; g < 3; ++g) {
				const uint32_t T = 3 * j + g;
				const bool isFan = T + 2 < N;
				const uint32_t Fq = T + 2 - N; // wraps for fan triangles, unused there
				const uint32_t ed = Fq >> 1;
				const bool second = (Fq & 1u) != 0;
				const uint32_t fb = 2 * ed;
				const bool lastEdge = ed + 1 == N;
				const uint32_t nextInner = lastEdge ? 0u : fb + 2, nextOuter = lastEdge ? 1u : fb + 3;
				// + F.ibase: vertex-buffer relative when assembly is armed (uint16 wrap = the reference's cast, vg_util.cpp:447)
				val[3 * g] = ((isFan ? 0u : fb) + F.ibase) & 0xFFFFu;
				val[3 * g + 1] = ((isFan ? 2 * T + 2 : (second ? nextOuter : fb + 1)) + F.ibase) & 0xFFFFu;
				val[3 * g + 2] = ((isFan ? 2 * T + 4 : (second ? nextInner : nextOuter)) + F.ibase) & 0xFFFFu;
			}
			if (F.sseOrder) {
				// VGX_FILL_INDEX_ORDER_SSE (stroker.cpp:610-701): [quad 0] then per fan triangle t {(0, s, s+2), quad of edge t+1 =
				// (s, s+1, s+3, s, s+3, s+2)} with s = 2t+2, then the wrap-around quad (L, L+1, 1, L, 1, 0), L = 2N-2. Positions
				// [9j, 9j+9) are therefore: the quad of edge j, then fan triangle j -- or, for the last two corners, the halves of
				// the wrap-around quad.
				const uint32_t b = 2 * j;
				const bool lastFan = j + 2 >= N; // corner N-2: its three trailing positions start the wrap-around quad
				val[0] = b; val[1] = b + 1; val[2] = b + 3; val[3] = b; val[4] = b + 3; val[5] = b + 2;
				val[6] = lastFan ? b + 2 : 0u; val[7] = lastFan ? b + 3 : b + 2; val[8] = lastFan ? 1u : b + 4;
				if (j + 1 == N) { val[0] = b; val[1] = 1u; val[2] = 0u; } // corner N-1: (L, 1, 0)
#pragma unroll
				for (uint32_t g = 0; g < 9; ++g) { val[g] = (val[g] + F.ibase) & 0xFFFFu; }
			}
#endif
}

VGX_EL void fill_emit_store(float* pos, uint32_t* color_out, uint16_t* idx_out, const FillFetch& F, V2 dPrev, V2 d12);

__device__ __forceinline__ void fill_emit_chunk(float* pos, uint32_t* color_out, uint16_t* idx_out, const FillFetch& F)
{
	const V2 p1 = F.p1;
	V2 pNext;
	pNext.x = wave_from_next(p1.x, 0.0f); pNext.y = wave_from_next(p1.y, 0.0f);
	if (!F.nextInWave) { pNext = F.pNextB; }
	V2 d12 = v2(0.0f, 0.0f);
#ifdef VGX_EXP_CHEAPMATH
	if (F.aaElem) { d12 = v2sub(pNext, p1); }
#else
	if (F.aaElem) { d12 = v2dir(p1, pNext); }
#endif
	V2 dPrev;
	dPrev.x = wave_from_prev(d12.x, 0.0f); dPrev.y = wave_from_prev(d12.y, 0.0f);
#ifdef VGX_EXP_CHEAPMATH
	if (F.aaElem && !F.prevInWave) { dPrev = v2sub(p1, F.pPrevB); }
#else
	if (F.aaElem && !F.prevInWave) { dPrev = v2dir(F.pPrevB, p1); }
#endif
	fill_emit_store(pos, color_out, idx_out, F, dPrev, d12);
}

// The stores of one fill element whose two edge directions are known (dPrev = vec2Dir(previous corner, p1), d12 =
// vec2Dir(p1, next corner)); fill_emit_chunk gets them from the neighbouring lanes, the template emitter (vgx_tmpl.hip)
// computes them per lane.
VGX_EL void fill_emit_store(float* pos, uint32_t* color_out, uint16_t* idx_out, const FillFetch& F, V2 dPrev, V2 d12)
{
	const bool valid = F.valid;
	const uint32_t j = F.j, N = F.N, color = F.color;
	const V2 p1 = F.p1;
	if (valid) {
		if (F.aaElem) {
#ifdef VGX_EXP_CHEAPMATH
			const V2 vaa = v2mul(v2add(dPrev, d12), F.aa);
#else
			const V2 vaa = v2mul(v2extrude(dPrev, d12), F.aa);
#endif
			const V2 vin = v2add(p1, vaa), vout = v2sub(p1, vaa);
			const uint64_t gv = F.firstV + 2 * (uint64_t)j;
			PosPair pp; pp.x0 = vin.x; pp.y0 = vin.y; pp.x1 = vout.x; pp.y1 = vout.y;
#ifdef VGX_EXP_NOPOS
			if ((__float_as_uint(pp.x0) ^ __float_as_uint(pp.y1)) == 0x7FEDCBA9u)
#endif
			VGX_ST_GUARD(__float_as_uint(pp.x0) ^ __float_as_uint(pp.y1)) { *(PosPair*)(pos + 2 * gv) = pp; }
			ColPair cp; cp.c0 = color; cp.c1 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
#ifdef VGX_EXP_NOCOL
			if (cp.c0 == 0x7FEDCBA9u)
#endif
			VGX_ST_GUARD(cp.c0) { *(ColPair*)(color_out + gv) = cp; }
			// indices: my nine positions [9j, 9j+9) are three whole triangles T = 3j + g (the fan size 3(N-2) and the
			// fringe quads are multiples of 3): T < N-2 is fan triangle (0, 2T+2, 2T+4) (stroker.cpp:769-776), else
			// fringe triangle F = T-(N-2) = half (F&1) of the quad on edge F>>1: (fb, fb+1, nextOuter) /
			// (fb, nextOuter, nextInner) with fb = 2*edge (stroker.cpp:779-795). No division, no per-index select.
			const uint32_t k9 = 9 * j;
			uint32_t val[9];
			fill_idx9(j, N, F.ibase, F.sseOrder, val);
			uint16_t* pi = idx_out + F.firstI + k9;
			if (j + 1 < N) {
				Idx9 q; q.a = val[0] | (val[1] << 16); q.b = val[2] | (val[3] << 16); q.c = val[4] | (val[5] << 16); q.d = val[6] | (val[7] << 16); q.e = (uint16_t)val[8];
#ifdef VGX_EXP_NOIDX
				if ((q.a ^ q.b ^ q.c ^ q.d ^ q.e) == 0x7FEDCBA9u)
#endif
				VGX_ST_GUARD(q.a ^ q.b ^ q.c ^ q.d ^ q.e) { *(Idx9*)pi = q; }
			} else {
				Idx3 q; q.a = val[0] | (val[1] << 16); q.b = (uint16_t)val[2];
#ifdef VGX_EXP_NOIDX3
				if ((q.a ^ q.b) == 0x7FEDCBA9u)
#endif
				*(Idx3*)pi = q;
			}
		} else {
			const uint64_t gv = F.firstV + j;
			*(float2*)(pos + 2 * gv) = make_float2(p1.x, p1.y);
			color_out[gv] = color;
			if (j + 2 < N) {
				uint16_t* pi = idx_out + F.firstI + 3 * j;
				pi[0] = (uint16_t)F.ibase; pi[1] = (uint16_t)(j + 1 + F.ibase); pi[2] = (uint16_t)(j + 2 + F.ibase);
			}
		}
	}
}

} // namespace

#endif
