// vgx_concave_lane.h -- the per-vertex arithmetic of strokerConcaveFillEndAA's fringe loop (reference src/stroker.cpp:887-973) for ONE
// contour vertex: host + device. The kernels of vgx_concave.hip run it one contour vertex per lane; the host backend of the per-call
// compat layer (host/vgx_host_backend.hip) runs the same functions vertex after vertex.
#ifndef VGX_CONCAVE_LANE_H
#define VGX_CONCAVE_LANE_H

#include "vgx_lane.h"

VGX_HD V2 ldc(const float* v, uint64_t i)
{
	const float2 t = *(const float2*)(v + 2 * i);
	return v2(t.x, t.y);
}

struct FringePair { V2 in, out; }; // p[inner], p[1 - inner]

// contour-wide constants: aa = fringe / 2 * sign(cross(d(last,0), d(0,1))), inner = sign < 0 ? 0 : 1 (stroker.cpp:895-898)
VGX_HD float contour_cross_sign(const float* v, uint32_t n)
{
	const V2 d01 = v2dir(ldc(v, n - 1), ldc(v, 0));
	return vgm_sign(v2cross(d01, v2dir(ldc(v, 0), ldc(v, n > 1 ? 1 : 0))));
}

VGX_HD FringePair fringe_of(V2 p1, V2 d01, V2 d12, float aa, bool innerIsSecond)
{
	const V2 vaa = v2mul(v2extrude(d01, d12), aa);
	const V2 p0 = v2sub(p1, vaa), pp1 = v2add(p1, vaa);
	FringePair r;
	r.in = innerIsSecond ? pp1 : p0;
	r.out = innerIsSecond ? p0 : pp1;
	return r;
}

// vertex j of a contour of n original vertices v[0..n): both fringe vertices (stroker.cpp:899-927)
VGX_HD FringePair contour_vertex(const float* v, uint32_t n, uint32_t j, float fringe)
{
	const float crossSign = contour_cross_sign(v, n);
	const float aa = fringe * 0.5f * crossSign;
	const bool innerIsSecond = !(crossSign < 0.0f);
	const V2 p1 = ldc(v, j);
	const V2 pPrev = ldc(v, j > 0 ? j - 1 : n - 1);
	const V2 d01 = v2dir(pPrev, p1); // iteration j's d01 = iteration j-1's d12 = dir(original v[j-1], original v[j]); j = 0: dir(v[n-1], v[0])
	V2 p2 = ldc(v, j + 1 < n ? j + 1 : 0);
	if (j + 1 == n && n > 1) { // the closing iteration reads vertex 0 AFTER iteration 0 moved it
		const V2 q0 = ldc(v, 0);
		const FringePair f0 = fringe_of(q0, v2dir(ldc(v, n - 1), q0), v2dir(q0, ldc(v, 1)), aa, innerIsSecond);
		p2 = f0.in;
	}
	return fringe_of(p1, d01, v2dir(p1, p2), aa, innerIsSecond);
}

#endif
