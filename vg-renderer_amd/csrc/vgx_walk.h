// vgx_walk.h -- the per-lane adaptive cubic walk (pathCubicTo, reference src/path.cpp:86-182) with its pending
// stack in LDS, the leaf sinks and the lane-resident draw window: shared by the flatten kernels (vgx_flatten.hip) and
// the fused single-pass kernel (vgx_fused.hip). Device only.
#ifndef VGX_WALK_H
#define VGX_WALK_H

#include "vgx_internal.h"
#include "vgx_wave.h"

namespace {

// ---- pending stack of the cubic DFS: the first VGX_LDS_LEVELS levels in LDS as [level][3 points][64 lanes]
// float2 (lane-interleaved -> conflict free for any mix of levels), deeper levels (only very fine subdivisions
// reach them) in per-lane private memory. Fewer LDS bytes per wave = more resident waves to hide latency.
#define VGX_LDS_LEVELS 4
template<int LV>
struct LdsLevelsT // the first VGX_LDS_LEVELS levels only (hot loop: no private-memory branch)
{
	float2* base; // &s_stack[lane]
	__device__ __forceinline__ void push(int level, float ax, float ay, float bx, float by, float cx, float cy)
	{
		float2* p = base + level * 3 * VGX_WAVE;
		p[0] = make_float2(ax, ay);
		p[VGX_WAVE] = make_float2(bx, by);
		p[2 * VGX_WAVE] = make_float2(cx, cy);
	}
	__device__ __forceinline__ void pop(int level, float& ax, float& ay, float& bx, float& by, float& cx, float& cy)
	{
		const float2* p = base + level * 3 * VGX_WAVE;
		const float2 a = p[0], b = p[VGX_WAVE], c = p[2 * VGX_WAVE];
		ax = a.x; ay = a.y; bx = b.x; by = b.y; cx = c.x; cy = c.y;
	}
};

template<int LV>
struct LdsStackT // full depth: LDS levels first, the rest in private memory
{
	float2* base; // &s_stack[lane]
	float deep[(VGX_CUBIC_MAX_PENDING - LV) * 6];
	__device__ __forceinline__ void push(int level, float ax, float ay, float bx, float by, float cx, float cy)
	{
		if (level < LV) {
			float2* p = base + level * 3 * VGX_WAVE;
			p[0] = make_float2(ax, ay);
			p[VGX_WAVE] = make_float2(bx, by);
			p[2 * VGX_WAVE] = make_float2(cx, cy);
		} else {
			float* p = deep + (level - LV) * 6;
			p[0] = ax; p[1] = ay; p[2] = bx; p[3] = by; p[4] = cx; p[5] = cy;
		}
	}
	__device__ __forceinline__ void pop(int level, float& ax, float& ay, float& bx, float& by, float& cx, float& cy)
	{
		if (level < LV) {
			const float2* p = base + level * 3 * VGX_WAVE;
			const float2 a = p[0], b = p[VGX_WAVE], c = p[2 * VGX_WAVE];
			ax = a.x; ay = a.y; bx = b.x; by = b.y; cx = c.x; cy = c.y;
		} else {
			const float* p = deep + (level - LV) * 6;
			ax = p[0]; ay = p[1]; bx = p[2]; by = p[3]; cx = p[4]; cy = p[5];
		}
	}
};

typedef LdsLevelsT<VGX_LDS_LEVELS> LdsLevels;
typedef LdsStackT<VGX_LDS_LEVELS> LdsStack;

// Common case on the LDS-only stack; a cubic that nests deeper than VGX_LDS_LEVELS pending halves is redone from its
// root with the full-depth stack (it gives up on its first too-deep descent, i.e. after a handful of steps).
template<int LV, class SINK>
__device__ __forceinline__ void wave_flatten_cubic(float x1, float y1, float x2, float y2, float x3, float y3, float x4, float y4, float tessTol, LdsStackT<LV>& stack, SINK& sink)
{
	const SINK fresh = sink;
	LdsLevelsT<LV> hot;
	hot.base = stack.base;
	if (vgx_flatten_cubic_n<LV, true>(x1, y1, x2, y2, x3, y3, x4, y4, tessTol, hot, sink)) {
		sink = fresh;
		vgx_flatten_cubic(x1, y1, x2, y2, x3, y3, x4, y4, tessTol, stack, sink);
	}
}

// ---- sink of the lane-parallel cubic: counts leaves, flags the cases that need the serial path -------
template<bool EMIT, bool XFORM>
struct FastCubicSink
{
	V2 prev;            // previous vertex of the polyline (for the epsilon test)
	uint32_t n;
	bool slow;
	// emit
	float* out;         // &poly[2 * first vertex of this command]
	uint32_t writeLimit;// vertices [0, writeLimit) are written (excludes a vertex popped by CLOSE)
	const float* mtx;   // state transform (used only when XFORM)
	__device__ __forceinline__ void leaf(float x, float y)
	{
		if (!EMIT) {
			slow = slow || v2near(prev, v2(x, y));
			prev = v2(x, y);
		} else if (n < writeLimit) {
			V2 p = v2(x, y);
			if (XFORM) { p = v2xform(p, mtx); }
			*(float2*)(out + 2 * (size_t)n) = make_float2(p.x, p.y);
		}
		++n;
	}
	__device__ __forceinline__ void dropped() { slow = true; }
};

__device__ __forceinline__ bool is_shape_cmd(uint32_t t) { return t >= VGX_CMD_RECT && t <= VGX_CMD_ELLIPSE; }

// Per-draw record held one per lane for a window of 64 consecutive draws (refilled when the walk leaves it): the
// command lanes get their draw's command base with a shuffle instead of a chain of dependent global loads.
struct DrawWindow
{
	uint64_t prefix; // cmd_prefix[wbase + lane] (or ~0 past the end)
	uint32_t pc0;    // first command of the draw's path
	uint32_t serial; // path must take the serial lane path (ARC / ARC_TO)
};

__device__ __forceinline__ DrawWindow draw_window_load(const VgxFlattenArgs& A, uint64_t wbase, int lane)
{
	DrawWindow w;
	const uint64_t idx = wbase + (uint64_t)lane;
	w.prefix = (idx <= A.ndraws) ? A.cmd_prefix[idx] : ~0ull;
	w.pc0 = 0; w.serial = 0;
	if (idx < A.ndraws) {
		const uint32_t path = A.draws[idx].path;
		w.pc0 = A.ps.path_cmd_begin[path];
		w.serial = A.ps.path_flags[path] & VGX_PF_SERIAL;
	}
	return w;
}

#define VGX_LEAF_SLOTS 8

struct BuildCubicSink // counts leaves, detects the serial-path cases, keeps the first leaves in the lane's LDS slots
{
	V2 prev;
	uint32_t n;
	bool slow;
	float2* slots; // &s_leaf[lane], stride VGX_WAVE
	float2* over;  // &overflow[lane], stride VGX_WAVE: leaves VGX_LEAF_SLOTS .. VGX_LEAF_SLOTS + VGX_BUILD_OVERFLOW - 1
	__device__ __forceinline__ void leaf(float x, float y)
	{
		slow = slow || v2near(prev, v2(x, y));
		prev = v2(x, y);
		if (n < VGX_LEAF_SLOTS) { slots[n * VGX_WAVE] = make_float2(x, y); }
		else if (n < VGX_LEAF_SLOTS + VGX_BUILD_OVERFLOW) { over[(n - VGX_LEAF_SLOTS) * VGX_WAVE] = make_float2(x, y); }
		++n;
	}
	__device__ __forceinline__ void dropped() { slow = true; }
};

// Hand-shaped hot loop of the build kernel: the same walk as vgx_flatten_cubic_n<VGX_LDS_LEVELS, true> + BuildCubicSink,
// with the points kept as packed float pairs (v_pk_add/mul_f32), running LDS addresses instead of level * stride
// multiplies, and the epsilon test folded into a running minimum. Arithmetic and its order are unchanged
// (path.cpp:107-170, 769-775). Returns false when the cubic nests deeper than the LDS levels (caller redoes it).
typedef float v2f __attribute__((ext_vector_type(2)));

template<int LV>
__device__ __forceinline__ bool build_flatten_hot(v2f P1, v2f P2, v2f P3, v2f P4, float tessTol, float2* stackLane, float2* slots, float2* over, uint32_t* nOut, bool* slowOut)
{
	int pending = 0;
	uint32_t n = 0;
	float minD2 = 3.0e38f; // smallest squared distance between consecutive vertices
	v2f prev = P1;
	bool more = true, aborted = false;
	uint32_t sp = 0; // next free stack entry, in float2 units relative to stackLane
	while (more) {
		const v2f d = P4 - P1;
		const v2f a2 = P2 - P4, a3 = P3 - P4;
		const v2f dsw = d.yx;
		const v2f m2 = a2 * dsw, m3 = a3 * dsw;
		const float d2 = __builtin_fabsf(m2.x - m2.y), d3 = __builtin_fabsf(m3.x - m3.y);
		const float d23 = d2 + d3;
		const v2f dd = d * d;
		const bool flat = d23 * d23 <= tessTol * (dd.x + dd.y);
		const bool push = !flat && pending < LV;
		const v2f P12 = (P1 + P2) * 0.5f, P23 = (P2 + P3) * 0.5f, P34 = (P3 + P4) * 0.5f;
		const v2f P123 = (P12 + P23) * 0.5f, P234 = (P23 + P34) * 0.5f;
		const v2f P1234 = (P123 + P234) * 0.5f;
		v2f N2 = P12, N3 = P123, N4 = P1234;
		if (push) {
			stackLane[sp] = make_float2(P234.x, P234.y);
			stackLane[sp + VGX_WAVE] = make_float2(P34.x, P34.y);
			stackLane[sp + 2 * VGX_WAVE] = make_float2(P4.x, P4.y);
			sp += 3 * VGX_WAVE;
		} else {
			if (flat) {
				const v2f e = prev - P4;
				const v2f ee = e * e;
				const float dist2 = ee.x + ee.y;
				minD2 = dist2 < minD2 ? dist2 : minD2;
				prev = P4;
				if (n < VGX_LEAF_SLOTS) { slots[n * VGX_WAVE] = make_float2(P4.x, P4.y); }
				else if (n < VGX_LEAF_SLOTS + VGX_BUILD_OVERFLOW) { over[(n - VGX_LEAF_SLOTS) * VGX_WAVE] = make_float2(P4.x, P4.y); }
				++n;
			} else {
				aborted = true;
			}
			P1 = P4;
			if (pending > 0) {
				sp -= 3 * VGX_WAVE;
				const float2 q2 = stackLane[sp], q3 = stackLane[sp + VGX_WAVE], q4 = stackLane[sp + 2 * VGX_WAVE];
				N2.x = q2.x; N2.y = q2.y; N3.x = q3.x; N3.y = q3.y; N4.x = q4.x; N4.y = q4.y;
			}
		}
		more = (push || pending > 0) && !aborted;
		pending += push ? 1 : -1;
		P2 = N2; P3 = N3; P4 = N4;
	}
	*nOut = n;
	*slowOut = minD2 < VGM_EPSILON;
	return !aborted;
}

} // namespace

#endif
