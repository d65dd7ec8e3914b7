// vgx_tmpl_elem.h -- the element routines the tile kernels share: one convex-fill element, one element of a closed Miter AA / Thin
// stroke, with direct wide stores at workgroup-uniform stream bases + 32-bit offsets. Used by the template kernels (vgx_tmpl.hip:
// instanced batches, vertices from the template through the instance's transform) and by the tile kernel of ordinary batches
// (vgx_tile.hip: vertices from the polyline heap). Device only.
#ifndef VGX_TMPL_ELEM_H
#define VGX_TMPL_ELEM_H

#include "vgx_elem.h"

namespace {

// Output streams of ONE instance: workgroup-uniform bases (scalar registers) + 32-bit byte offsets per lane, so that every
// store is `global_store saddr + voffset` instead of a 64-bit address built per lane (template mode requires an instance to
// stay below 4 GB per stream).
struct TmplOut { char* pos; char* col; char* idx; };
#ifdef VGX_EXP_NOIDX /* tuning experiment: the index stream is not stored (wrong output) */
#define TMPL_IDX_ON if (O.idx == nullptr)
#else
#define TMPL_IDX_ON
#endif

// One convex-fill element: strokerConvexFill / strokerConvexFillAA (stroker.cpp:334-365, 713-807), as fill_emit_store (vgx_elem.h).
__device__ __forceinline__ void tmpl_fill_elem(const TmplOut& O, uint32_t kindWord, uint32_t N, uint32_t vOff, uint32_t iOff, uint32_t ibase, uint32_t color, float aa,
	uint32_t j, V2 p1, V2 dPrev, V2 d12)
{
	if (VGX_MD_KIND(kindWord) == VGX_MESH_FILL_AA) {
		const V2 vaa = v2mul(v2extrude(dPrev, d12), aa);
		const V2 vin = v2add(p1, vaa), vout = v2sub(p1, vaa);
		const uint32_t gv = vOff + 2u * j;
		PosPair pp; pp.x0 = vin.x; pp.y0 = vin.y; pp.x1 = vout.x; pp.y1 = vout.y;
		ColPair cp; cp.c0 = color; cp.c1 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
		if (VGX_MD_SSE_ORDER(kindWord) == 0) {
			// Scalar index order (stroker.cpp:769-795): [N - 2 fan triangles (0, 2t + 2, 2t + 4)][N fringe quads, edge e:
			// (2e, 2e + 1, nextOuter) (2e, nextOuter, nextInner)]. Corner j writes ITS OWN fan triangle (j < N - 2) and the quad of its
			// own edge -- two short runs per lane, both lane-consecutive in memory, and a dozen integer operations, where a
			// contiguous nine-index slice per corner (fill_idx9, what k_fill writes) costs ~75: the same bytes at the same places.
			const uint32_t b2 = 2u * j + ibase; // + ibase: command relative when assembly is armed (uint16 wrap = the reference's cast, vg_util.cpp:447)
			const uint32_t ni = (j + 1 == N) ? ibase : b2 + 2u, no = ni + 1u;
			Idx6 quad; quad.a = (b2 & 0xFFFFu) | ((b2 + 1u) << 16); quad.b = (no & 0xFFFFu) | (b2 << 16); quad.c = (no & 0xFFFFu) | (ni << 16);
			*(PosPair*)(O.pos + gv * 8u) = pp;
			*(ColPair*)(O.col + gv * 4u) = cp;
			TMPL_IDX_ON *(Idx6*)(O.idx + (iOff + 3u * (N - 2u) + 6u * j) * 2u) = quad;
			if (j + 2 < N) {
				Idx3 fan; fan.a = (ibase & 0xFFFFu) | ((b2 + 2u) << 16); fan.b = (uint16_t)(b2 + 4u);
				TMPL_IDX_ON *(Idx3*)(O.idx + (iOff + 3u * j) * 2u) = fan;
			}
			return;
		}
		const uint32_t ib = (iOff + 9u * j) * 2u;
		uint32_t val[9];
		fill_idx9(j, N, ibase, true, val); // VGX_FILL_INDEX_ORDER_SSE: the quad of edge j, then fan triangle j -- contiguous per corner by itself
		VGX_ST_GUARD(cp.c0 ^ __float_as_uint(pp.x0)) {
		*(PosPair*)(O.pos + gv * 8u) = pp;
		*(ColPair*)(O.col + gv * 4u) = cp;
		if (j + 1 < N) {
			Idx9 q; q.a = val[0] | (val[1] << 16); q.b = val[2] | (val[3] << 16); q.c = val[4] | (val[5] << 16); q.d = val[6] | (val[7] << 16); q.e = (uint16_t)val[8];
			TMPL_IDX_ON *(Idx9*)(O.idx + ib) = q;
		} else {
			Idx3 q; q.a = val[0] | (val[1] << 16); q.b = (uint16_t)val[2];
			TMPL_IDX_ON *(Idx3*)(O.idx + ib) = q;
		}
		}
	} else {
		const uint32_t gv = vOff + j;
		VGX_ST_GUARD(color) {
		*(float2*)(O.pos + gv * 8u) = make_float2(p1.x, p1.y);
		*(uint32_t*)(O.col + gv * 4u) = color;
		if (j + 2 < N) { // fan (0, j + 1, j + 2), stroker.cpp:340-357
			Idx3 q; q.a = (ibase & 0xFFFFu) | ((j + 1 + ibase) << 16); q.b = (uint16_t)(j + 2 + ibase);
			TMPL_IDX_ON *(Idx3*)(O.idx + (iOff + 3u * j) * 2u) = q;
		}
		}
	}
}

// One element of a CLOSED stroke with MITER joins, AA (4 rails) or Thin (3 rails): stroke_chunk_simple (vgx_elem.h) without
// its neighbour lanes. dPrev2 = direction of the edge in front of the (cyclically) previous vertex (that join's inner side is
// recomputed from it): same
// inputs, same arithmetic, same bits as the values the sequential stroker carries along (stroker.cpp:1401-1410).
__device__ __forceinline__ void tmpl_stroke_elem(const TmplOut& O, uint32_t kindWord, uint32_t N, uint32_t vOff, uint32_t iOff, uint32_t ibase, uint32_t color, float hsw, float hswAA,
	uint32_t j, V2 p1, V2 dPrev2, V2 dPrev, V2 d12)
{
	const bool thin = VGX_MD_KIND(kindWord) == VGX_MESH_STROKE_AA_THIN;
	const uint32_t R = thin ? 3u : 4u;
	const uint32_t bridgeIdx = thin ? 12u : 18u;
	const float sideWidth = thin ? hsw : hswAA; // fringe : hswAA
	const VgxJoin jn = vgx_join_dirs(dPrev, d12, sideWidth);
	const bool L = jn.leftInner;
	const uint32_t b = R * j;
	const uint32_t bi = b + ibase, top = bi + R - 1; // index VALUES carry the assembly base, positions in the streams do not
	const Rails mine = thin ? (L ? rails(bi, bi + 1, bi + 2, 0) : rails(top, bi + 1, bi, 0)) : (L ? rails(bi, bi + 1, bi + 2, bi + 3) : rails(top, bi + 2, bi + 1, bi));
	const uint32_t c0 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
	char* pp = O.pos + (vOff + b) * 8u;
	char* pc = O.col + (vOff + b) * 4u;
	if (thin) { // stroker.cpp:2060-2110
		const V2 vf = v2mul(jn.v, hsw);
		const V2 q0 = L ? v2add(p1, vf) : v2sub(p1, vf);
		const V2 q2 = L ? v2sub(p1, vf) : v2add(p1, vf);
		PosPair q; q.x0 = q0.x; q.y0 = q0.y; q.x1 = p1.x; q.y1 = p1.y;
		ColPair c; c.c0 = c0; c.c1 = color;
		VGX_ST_GUARD(c0) {
		*(PosPair*)pp = q;
		*(float2*)(pp + 16) = make_float2(q2.x, q2.y);
		*(ColPair*)pc = c;
		*(uint32_t*)(pc + 8) = c0;
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
		struct __attribute__((packed, aligned(4))) ColQuad { uint32_t a, b, c, d; };
		ColQuad cq; cq.a = c0; cq.b = color; cq.c = color; cq.d = c0; // ONE 16-byte store: consecutive lanes, consecutive 16 bytes (two 8-byte stores at a 16-byte stride made every line a target of two instructions = two write requests of half a line each)
		VGX_ST_GUARD(c0 ^ __float_as_uint(q.x0) ^ __float_as_uint(r.y1)) {
		*(PosPair*)pp = q;
		*(PosPair*)(pp + 16) = r;
#ifdef VGX_EXP_COL2X8
		ColPair c; c.c0 = c0; c.c1 = color;
		ColPair d; d.c0 = color; d.c1 = c0;
		*(ColPair*)pc = c;
		*(ColPair*)(pc + 8) = d;
#else
		*(ColQuad*)pc = cq;
#endif
		}
	}
	{
		// The bridge that ENDS at this join: from join j - 1 (stroker.cpp:1557-1564, 1714-1721; thin :2093-2098, 2175-2180) or,
		// for join 0, from the LAST join -- the closing bridge, which the reference appends behind the last join's own bridge
		// (:1970-1984, 2295-2306: prevSegment = the last join's rails, first = join 0's: the same six triangles as any bridge).
		// Every element so writes exactly one bridge (element 0 at the END of the mesh's index range) and recomputes exactly
		// one neighbouring join's inner side, instead of the last element doing two of each.
		const uint32_t jm = j > 0 ? j - 1 : N - 1;
		const VgxJoin jp = vgx_join_dirs(dPrev2, dPrev, sideWidth);
		const uint32_t pb = R * jm + ibase, ptop = pb + R - 1;
		const Rails p = thin ? (jp.leftInner ? rails(pb, pb + 1, pb + 2, 0) : rails(ptop, pb + 1, pb, 0))
		                     : (jp.leftInner ? rails(pb, pb + 1, pb + 2, pb + 3) : rails(ptop, pb + 2, pb + 1, pb));
		char* pi = O.idx + (iOff + bridgeIdx * jm) * 2u;
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

} // namespace

#endif
