// vgx_lane.h -- per-lane geometry used by the HIP kernels (flatten + stroker).
//
// Everything here is what ONE lane does for ONE path command or ONE polyline vertex; the kernels in
// vgx_flatten.hip / vgx_stroke.hip add the wavefront-level parts (ballot, prefix scans, offsets).
// Functions are __host__ __device__ so the same arithmetic can be unit-tested on the CPU
// (csrc/vgx_hosttest.cpp); the product path only ever runs them on the device.
//
// Arithmetic contract: binary32, evaluation order exactly as in the cited reference lines, no FMA
// contraction (-ffp-contract=off), transcendentals from vgmath.h.
#ifndef VGX_LANE_H
#define VGX_LANE_H

#include "vgmath.h"
#include "../../include/vgx.h"
#if defined(__HIPCC__) || defined(__HIP__)
#include "vgx_fastmath.h"
#endif

#if defined(__HIPCC__) || defined(__HIP__)
#define VGX_HD __host__ __device__ __forceinline__
#define VGX_HDM __host__ __device__ __forceinline__ /* member functions */
#else
#define VGX_HD static inline
#define VGX_HDM inline
#endif

struct V2 { float x, y; };

VGX_HD V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
VGX_HD V2 v2add(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
VGX_HD V2 v2sub(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
VGX_HD V2 v2mul(V2 a, float s) { return v2(a.x * s, a.y * s); }
VGX_HD V2 v2ccw(V2 a) { return v2(-a.y, a.x); }
VGX_HD V2 v2cw(V2 a) { return v2(a.y, -a.x); }
VGX_HD float v2dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
VGX_HD float v2cross(V2 a, V2 b) { return a.x * b.y - b.x * a.y; }

// vec2Dir, stroker.cpp:31-38
// Device code takes 1 / sqrt(lenSqr) from vgx_fastmath.h: the same correctly rounded value as vgm_rsqrt in a third of the
// instructions (checked over every float of its domain by tests/native/exact_math_test.hip); values outside the checked
// domain -- coordinates beyond 1e15, NaN / Inf -- take the generic sequence under a branch that never runs.
VGX_HD V2 v2dir(V2 a, V2 b)
{
	const float dx = b.x - a.x;
	const float dy = b.y - a.y;
	const float lenSqr = dx * dx + dy * dy;
#if defined(__HIP_DEVICE_COMPILE__)
	float inv = vgx_rsqrt_rn(lenSqr);
	if (__builtin_expect(!(lenSqr <= 0x1p100f), 0)) { inv = vgm_rsqrt(lenSqr); }
	const float invLen = lenSqr < VGM_EPSILON ? 0.0f : inv;
#else
	const float invLen = lenSqr < VGM_EPSILON ? 0.0f : vgm_rsqrt(lenSqr);
#endif
	return v2(dx * invLen, dy * invLen);
}

// calcExtrusionVector, stroker.cpp:40-53
VGX_HD V2 v2extrude(V2 d01, V2 d12)
{
	V2 v = v2ccw(d01);
	const float c = v2cross(d12, d01);
	const float ac = vgm_abs(c);
	if (ac > (1.0f / 100.0f)) {
#if defined(__HIP_DEVICE_COMPILE__)
		float r = vgx_rcp_rn(c);
		if (__builtin_expect(!(ac <= 0x1p100f), 0)) { r = 1.0f / c; }
#else
		const float r = 1.0f / c;
#endif
		v = v2mul(v2sub(d01, d12), r);
	}
	return v;
}

// transformPos2D, vg_util.h:24-28 (scalar association (m0*x + m2*y) + m4)
VGX_HD V2 v2xform(V2 p, const float* m)
{
	return v2(m[0] * p.x + m[2] * p.y + m[4], m[1] * p.x + m[3] * p.y + m[5]);
}

VGX_HD bool v2near(V2 a, V2 b) // the epsilon test of pathAddVertex / pathClose (path.cpp:769-775, 718-722)
{
	const float dx = a.x - b.x;
	const float dy = a.y - b.y;
	return dx * dx + dy * dy < VGM_EPSILON;
}

// ------------------------------------------------------------------------------------------------
// Adaptive cubic flattening (pathCubicTo, path.cpp:86-182) as a per-lane depth-first walk.
//
// STACK keeps the pending right halves (at most 10, path.cpp:90). Only three points per entry are
// stored: the right half's first point is always the current piece's end point at pop time.
// SINK receives leaves in curve order: sink.leaf(x, y); sink.dropped() is called when a piece is
// discarded because the stack is full (path.cpp:168-179).
// ------------------------------------------------------------------------------------------------
#define VGX_CUBIC_MAX_PENDING 10

// vgx_flatten_cubic_n<LIMIT, ABORT>: LIMIT = capacity of STACK. With ABORT the walk gives up (returns true, the
// sink is left in an undefined state) the first time a push would exceed LIMIT; callers then redo the cubic with the
// full-depth stack. That lets the wave-parallel kernels run their common case on a small LDS-only stack.
template<int LIMIT, bool ABORT, class STACK, class SINK>
VGX_HD bool vgx_flatten_cubic_n(float x1, float y1, float x2, float y2, float x3, float y3, float x4, float y4, float tessTol, STACK& stack, SINK& sink)
{
	// One loop exit and select-style state updates: the wave runs this loop in lockstep, so every extra branch target
	// costs exec-mask bookkeeping on ALL lanes. Same arithmetic, same order as path.cpp:107-170.
	int pending = 0;
	bool more = true, aborted = false;
	while (more) {
		const float dx = x4 - x1;
		const float dy = y4 - y1;
		const float d2 = vgm_abs((x2 - x4) * dy - (y2 - y4) * dx);
		const float d3 = vgm_abs((x3 - x4) * dy - (y3 - y4) * dx);
		const float d23 = d2 + d3;
		const bool flat = d23 * d23 <= tessTol * (dx * dx + dy * dy);
		const bool push = !flat && pending < LIMIT;
		const float x12 = (x1 + x2) * 0.5f, y12 = (y1 + y2) * 0.5f;
		const float x23 = (x2 + x3) * 0.5f, y23 = (y2 + y3) * 0.5f;
		const float x34 = (x3 + x4) * 0.5f, y34 = (y3 + y4) * 0.5f;
		const float x123 = (x12 + x23) * 0.5f, y123 = (y12 + y23) * 0.5f;
		const float x234 = (x23 + x34) * 0.5f, y234 = (y23 + y34) * 0.5f;
		const float x1234 = (x123 + x234) * 0.5f, y1234 = (y123 + y234) * 0.5f;
		float nx2 = x12, ny2 = y12, nx3 = x123, ny3 = y123, nx4 = x1234, ny4 = y1234;
		if (push) {
			stack.push(pending, x234, y234, x34, y34, x4, y4); // the right half waits; descend into the left half
		} else {
			if (flat) { sink.leaf(x4, y4); }
			else if (ABORT) { aborted = true; }
			else { sink.dropped(); } // stack full: the node is silently skipped
			x1 = x4; y1 = y4; // the next sibling starts where this subtree ended
			if (pending > 0) { stack.pop(pending - 1, nx2, ny2, nx3, ny3, nx4, ny4); }
		}
		more = (push || pending > 0) && !aborted;
		pending += push ? 1 : -1;
		x2 = nx2; y2 = ny2; x3 = nx3; y3 = ny3; x4 = nx4; y4 = ny4;
	}
	return aborted;
}

template<class STACK, class SINK>
VGX_HD void vgx_flatten_cubic(float x1, float y1, float x2, float y2, float x3, float y3, float x4, float y4, float tessTol, STACK& stack, SINK& sink)
{
	(void)vgx_flatten_cubic_n<VGX_CUBIC_MAX_PENDING, false>(x1, y1, x2, y2, x3, y3, x4, y4, tessTol, stack, sink);
}

// Quadratic -> cubic control points (pathQuadraticTo, path.cpp:184-201)
VGX_HD void vgx_quad_to_cubic(float x0, float y0, float cx, float cy, float x, float y, float* c1x, float* c1y, float* c2x, float* c2y)
{
	*c1x = x0 + (2.0f / 3.0f) * (cx - x0);
	*c1y = y0 + (2.0f / 3.0f) * (cy - y0);
	*c2x = x + (2.0f / 3.0f) * (cx - x);
	*c2y = y + (2.0f / 3.0f) * (cy - y);
}

// da = 2*acos(s*r/(s*r+tol)) (path.cpp:307,602,654; stroker.cpp:1013,1398)
VGX_HD float vgx_step_angle(float scale, float r, float tol)
{
	return vgm_acos((scale * r) / ((scale * r) + tol)) * 2.0f;
}

// float -> point count. The reference casts with (uint32_t) (stroker.cpp:1014, 1146; path.cpp:307, 655): undefined for
// Inf / NaN / >= 2^32, which is what a radius beyond ~4e6 tolerances produces (acos(1) = 0 -> pi / 0). Here the
// count saturates at VGX_MAX_ARC_POINTS: more than any valid mesh can hold (65536 vertices, vg.cpp:734), so such a
// shape ends as VGX_E_MESH_TOO_LARGE after a bounded amount of work instead of a 4-billion-step loop. NaN -> 0.
#define VGX_MAX_ARC_POINTS 131072u
VGX_HD uint32_t vgx_point_count(float x)
{
	return (x >= (float)VGX_MAX_ARC_POINTS) ? VGX_MAX_ARC_POINTS : ((x > 0.0f) ? (uint32_t)x : 0u);
}

VGX_HD uint32_t vgx_half_circle_points(float da) // max(2, ceil(pi/da))
{
	return vgm_umax(2u, vgx_point_count(vgm_ceil(VGM_PI / da)));
}

// ------------------------------------------------------------------------------------------------
// Stroker join geometry shared by the three polyline strokers.
// ------------------------------------------------------------------------------------------------
struct VgxJoin
{
	V2 d01, d12, v; // unit directions of the two segments, unit-width extrusion vector
	bool leftInner; // dot(d12, v*w) >= 0 (stroker.cpp:1099-1100, 1534-1535, 2072-2073)
};

VGX_HD VgxJoin vgx_join_dirs(V2 d01, V2 d12, float sideWidth)
{
	VgxJoin j;
	j.d01 = d01;
	j.d12 = d12;
	j.v = v2extrude(j.d01, j.d12);
	const V2 vw = v2mul(j.v, sideWidth);
	j.leftInner = (j.d12.x * vw.x + j.d12.y * vw.y) >= 0.0f;
	return j;
}

VGX_HD VgxJoin vgx_join(V2 p0, V2 p1, V2 p2, float sideWidth)
{
	VgxJoin j;
	j.d01 = v2dir(p0, p1);
	j.d12 = v2dir(p1, p2);
	j.v = v2extrude(j.d01, j.d12);
	const V2 vw = v2mul(j.v, sideWidth);
	j.leftInner = (j.d12.x * vw.x + j.d12.y * vw.y) >= 0.0f;
	return j;
}

struct VgxArc { float a01, arcDa; uint32_t n; };

// Round-join arc (stroker.cpp:1140-1147 / 1238-1245 / 1588-1595 / 1744-1751). n01/n12 are the
// outer-side normals (perpCW for a left-inner join, perpCCW for a right-inner one).
VGX_HD VgxArc vgx_round_join_arc(V2 n01, V2 n12, bool leftInner, float da)
{
	VgxArc r;
	const float a01 = vgm_atan2(n01.y, n01.x);
	float a12 = vgm_atan2(n12.y, n12.x);
	if (leftInner) {
		if (a12 < a01) { a12 += VGM_PI2; }
		r.n = vgm_umax(2u, vgx_point_count((a12 - a01) / da));
	} else {
		if (a12 > a01) { a12 -= VGM_PI2; }
		r.n = vgm_umax(2u, vgx_point_count((a01 - a12) / da));
	}
	r.a01 = a01;
	r.arcDa = (a12 - a01) / (float)r.n;
	return r;
}

// ------------------------------------------------------------------------------------------------
// Closed-form mesh sizes. Every stroker entry point except Round JOINS produces a vertex / index count
// that depends only on the polyline length N, closed flag, cap, join and (Round caps) the half-circle
// point count H -- the sums of the per-element counts listed in SURVEY.md 8a / appendix B.
// Returns false when the mesh contains Round joins (numArcPoints is data dependent, stroker.cpp:1146).
// ------------------------------------------------------------------------------------------------
// Sizes are computed in 64 bits and saturate (vertices at 0xFFFFFFFE -- above the 65536 a mesh may hold, so the
// scan over the meshes reports VGX_E_MESH_TOO_LARGE -- indices at 0xFFFFFFFF): a hostile draw cannot wrap a count
// back into the valid range (N up to 2^32-16 polyline vertices, H up to VGX_MAX_ARC_POINTS).
VGX_HD uint32_t vgx_sat_nv(uint64_t v) { return v > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)v; }
VGX_HD uint32_t vgx_sat_ni(uint64_t v) { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; }

VGX_HD bool vgx_mesh_closed_form(uint32_t kind, bool closed, uint32_t cap, uint32_t join, uint32_t N32, uint32_t H32, uint32_t* nv, uint32_t* ni)
{
	const uint64_t N = N32, H = H32;
	uint64_t v = 0, i = 0;
	if (kind == VGX_MESH_FILL) { v = N; i = 3 * (N - 2); }                            // stroker.cpp:336-337
	else if (kind == VGX_MESH_FILL_AA) { v = 2 * N; i = 9 * N - 6; }                  // stroker.cpp:728-732
	else if (kind == VGX_MESH_STROKE_AA_THIN) {
		const bool bevel = join != VGX_JOIN_MITER; // Round -> Bevel, stroker.cpp:318-327
		if (closed) { v = bevel ? 4 * N : 3 * N; i = bevel ? 15 * N : 12 * N; }
		else { v = bevel ? 4 * N - 2 : 3 * N; i = bevel ? 15 * N - 18 : 12 * (N - 1); }
	} else {
		if (join == VGX_JOIN_ROUND) { return false; }
		const bool bevel = join == VGX_JOIN_BEVEL;
		const bool roundCap = !closed && cap == VGX_CAP_ROUND;
		if (kind == VGX_MESH_STROKE_AA) {
			if (closed) { v = bevel ? 6 * N : 4 * N; i = bevel ? 27 * N : 18 * N; }
			else {
				const uint64_t joins = N - 2;
				const uint64_t capV = roundCap ? 4 * H : 8;
				const uint64_t capI = roundCap ? (9 * H - 12) + (18 + 3 * (H - 2) + 6 * (H - 1)) : 30;
				v = capV + joins * (bevel ? 6 : 4);
				i = capI + joins * (bevel ? 27 : 18);
			}
		} else { // VGX_MESH_STROKE
			if (closed) { v = bevel ? 3 * N : 2 * N; i = bevel ? 9 * N : 6 * N; }
			else {
				const uint64_t joins = N - 2;
				const uint64_t capV = roundCap ? 2 * H : 4;
				const uint64_t capI = roundCap ? 3 * (H - 2) + 6 + 3 * (H - 2) : 6;
				v = capV + joins * (bevel ? 3 : 2);
				i = capI + joins * (bevel ? 9 : 6);
			}
		}
	}
	*nv = vgx_sat_nv(v);
	*ni = vgx_sat_ni(i);
	return true;
}

// Stroker parameters of one mesh derived from its draw record (the values the reference computes at the top
// of polylineStroke / polylineStrokeAA / polylineStrokeAAThin, stroker.cpp:1011-1014, 1396-1399, 1999).
struct VgxStrokeParams { uint32_t cap, join; float hsw, hswAA; };

VGX_HD VgxStrokeParams vgx_stroke_params(uint32_t kind, bool closed, uint32_t strokeFlags, float strokeWidth, float fringe, float scale, float tol)
{
	VgxStrokeParams p;
	p.cap = VGX_STROKE_CAP(strokeFlags);
	p.join = VGX_STROKE_JOIN(strokeFlags);
	if (closed) { p.cap = VGX_CAP_BUTT; } // closed strokes ignore the cap (dispatch tables stroker.cpp:246-268)
	if (kind == VGX_MESH_STROKE_AA) {
		p.hsw = (strokeWidth - fringe) * 0.5f;
		p.hswAA = p.hsw + fringe;
	} else if (kind == VGX_MESH_STROKE) {
		p.hsw = strokeWidth * 0.5f;
		p.hswAA = p.hsw;
	} else {
		if (p.cap == VGX_CAP_ROUND) { p.cap = VGX_CAP_SQUARE; }
		if (p.join == VGX_JOIN_ROUND) { p.join = VGX_JOIN_BEVEL; }
		p.hsw = fringe;
		p.hswAA = fringe;
	}
	(void)scale; (void)tol; // da = vgx_step_angle(scale, hsw, tol) is only needed for Round caps / joins: callers compute it lazily
	return p;
}

#endif // VGX_LANE_H
