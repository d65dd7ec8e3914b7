// vgx_walk.h -- the per-lane adaptive cubic walk (pathCubicTo, reference src/path.cpp:86-182) with its pending
// stack in LDS, the leaf sinks and the lane-resident draw window of the flatten kernels (vgx_flatten.hip). Device only.
#ifndef VGX_WALK_H
#define VGX_WALK_H

#include "vgx_internal.h"
#include "vgx_wave.h"

namespace {

// ---- pending stack of the cubic DFS: the first VGX_LDS_LEVELS levels in LDS as [level][3 points][64 lanes]
// float2 (lane-interleaved -> conflict free for any mix of levels), deeper levels (only very fine subdivisions
// reach them) in per-lane private memory. Fewer LDS bytes per wave = more resident waves to hide latency.
#ifndef VGX_LDS_LEVELS
#define VGX_LDS_LEVELS 4
#endif
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
struct __attribute__((packed, aligned(8))) WalkQuad { float x0, y0, x1, y1; }; // two vertices in one (element-aligned) 16-byte store

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
	// The leaves of one cubic leave in groups of FOUR (32 bytes: two 16-byte stores back to back): a lane's single-vertex stores are
	// 8 bytes apart in time and 64 lanes x ~360 bytes apart in space, so they reached HBM as partial sectors (PMC WRITE_SIZE on
	// 1 M cubics: 1.34 GB for 0.37 GB of vertices; groups of four: 0.94 GB, same kernel time). Measured and not kept: groups aligned
	// to 32 bytes in memory (0.72 GB, but +8 % time for the bookkeeping of the cubic's first, partial group); groups of eight with
	// the parked vertices in an indexed array (the array went to scratch: 3x the time). flush() after the walk writes the rest.
	V2 pend0, pend1, pend2;
	__device__ __forceinline__ void begin() {}
	__device__ __forceinline__ void leaf(float x, float y)
	{
		if (!EMIT) {
			slow = slow || v2near(prev, v2(x, y));
			prev = v2(x, y);
		} else if (n < writeLimit) {
			V2 p = v2(x, y);
			if (XFORM) { p = v2xform(p, mtx); }
			const uint32_t k = n & 3u;
			if (k == 3u) {
				WalkQuad a, b;
				a.x0 = pend0.x; a.y0 = pend0.y; a.x1 = pend1.x; a.y1 = pend1.y;
				b.x0 = pend2.x; b.y0 = pend2.y; b.x1 = p.x; b.y1 = p.y;
				float* o = out + 2 * (size_t)(n - 3u);
				*(WalkQuad*)o = a;
				*(WalkQuad*)(o + 4) = b;
			} else {
				pend0 = k == 0u ? p : pend0;
				pend1 = k == 1u ? p : pend1;
				pend2 = k == 2u ? p : pend2;
			}
		}
		++n;
	}
	__device__ __forceinline__ void flush() // after the walk (emit sinks): the vertices of the last, incomplete group
	{
		if (EMIT) {
			const uint32_t wl = n < writeLimit ? n : writeLimit;
			const uint32_t k = wl & 3u;
			float* o = out + 2 * (size_t)(wl - k);
			if (k >= 1u) { *(float2*)o = make_float2(pend0.x, pend0.y); }
			if (k >= 2u) { *(float2*)(o + 2) = make_float2(pend1.x, pend1.y); }
			if (k == 3u) { *(float2*)(o + 4) = make_float2(pend2.x, pend2.y); }
		}
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
	uint32_t serial; // bit 0: path must take the serial lane path (ARC / ARC_TO / shapes); bit 1: VGX_PF_THIN
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
		w.serial = A.ps.path_flags[path] & (VGX_PF_SERIAL | VGX_PF_THIN); // bit 0: serial lane path; bit 1: thin records
	}
	return w;
}

#ifndef VGX_LEAF_SLOTS
#define VGX_LEAF_SLOTS 8
#endif

template<int SLOTS>
struct BuildCubicSinkT // counts leaves, detects the serial-path cases, keeps the first leaves in the lane's LDS slots
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
		if (n < (uint32_t)SLOTS) { slots[n * VGX_WAVE] = make_float2(x, y); }
		else if (n < (uint32_t)SLOTS + VGX_BUILD_OVERFLOW) { over[(n - SLOTS) * VGX_WAVE] = make_float2(x, y); }
		++n;
	}
	__device__ __forceinline__ void dropped() { slow = true; }
};
typedef BuildCubicSinkT<VGX_LEAF_SLOTS> BuildCubicSink;

// Hand-shaped hot loop of the build kernel: the same walk as vgx_flatten_cubic_n<VGX_LDS_LEVELS, true> + BuildCubicSink,
// with the points kept as packed float pairs (v_pk_add/mul_f32), running LDS addresses instead of level * stride
// multiplies, and the epsilon test folded into a running minimum. Arithmetic and its order are unchanged
// (path.cpp:107-170, 769-775). Returns false when the cubic nests deeper than the LDS levels (caller redoes it).
typedef float v2f __attribute__((ext_vector_type(2)));

template<int LV, int SLOTS = VGX_LEAF_SLOTS>
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

// ------------------------------------------------------------------------------------------------
// Pooled cubic walk: the wave subdivides ALL cubics of a 64-command chunk together.
//
// The per-lane walk above runs in lock-step: a chunk costs as many iterations as its deepest cubic needs while most
// lanes (MOVE_TO / LINE_TO / CLOSE commands, one-segment cubics) idle -- 31 % lane use on the Tiger workload, ~100
// instructions per iteration. Here every curve piece is a TASK on a LIFO in LDS: a round pops up to 64 tasks (one per
// lane), tests flatness (path.cpp:105-116, same expressions in the same order), appends the flat pieces' end points to
// a leaf list and pushes the two halves of the others, all at positions given by ballot + popcount (the wavefront-
// level compaction of north_star). Rounds are full except at the tail of a chunk; the subdivision arithmetic is
// untouched, so the leaves are bit-identical.
//
// Order: a piece at depth d reached by the left/right decisions `path` (d bits) covers the dyadic interval
// [path << (MAXD - d), (path + 1) << (MAXD - d)) of its cubic; a leaf sets bit (path << (MAXD - d)) in its command's
// 32-bit mask (ds_or). Afterwards a command has popcount(mask) vertices and the rank of a leaf is the popcount of the
// mask below its bit: pool_place writes every listed leaf, transformed, straight to its final place (the owner lane's
// base / limit / transform come through shuffles). No per-lane leaf slots, no global overflow area, no copy loop.
// Off the pooled path (per-lane walk, two passes: count, then emit to the final place): cubics that are still not flat
// at depth VGX_POOL_MAXD, and whatever does not fit the task LIFO or the leaf list. pathAddVertex's epsilon test
// (path.cpp:769-775) compares a leaf with the previous vertex, which is the piece's own first point.
// ------------------------------------------------------------------------------------------------
#define VGX_POOL_CAP 160      /* tasks; the first 6144 bytes double as the per-lane walk's LDS stack */
#define VGX_POOL_LEAVES 224   /* leaf list entries */
#define VGX_POOL_MAXD 5
#define VGX_POOL_F_DEEP 1u
#define VGX_POOL_F_SLOW 2u

struct __attribute__((aligned(8))) PoolTaskTail { uint32_t meta; float tol; }; // owner lane | depth << 6 | path << 9

struct PoolLds // the wave's pool memory (VGX_POOL_BYTES of LDS, 16-byte aligned)
{
	float4* task;       // [CAP][2] {P1, P2} {P3, P4}
	PoolTaskTail* tail; // [CAP]
	float2* leaf;       // [LEAVES] end point of a flat piece (untransformed)
	uint32_t* leafMeta; // [LEAVES] owner lane | start << 6
	uint32_t* mask;     // [64] leaf-start bits of the lane's cubic
	uint32_t* flags;    // [64] VGX_POOL_F_*
};
#define VGX_POOL_BYTES (VGX_POOL_CAP * 40 + VGX_POOL_LEAVES * 12 + 2 * 256)

__device__ __forceinline__ PoolLds pool_carve(unsigned char* mem)
{
	PoolLds L;
	L.task = (float4*)mem;
	L.tail = (PoolTaskTail*)(mem + VGX_POOL_CAP * 32);
	L.leaf = (float2*)(mem + VGX_POOL_CAP * 40);
	L.leafMeta = (uint32_t*)(mem + VGX_POOL_CAP * 40 + VGX_POOL_LEAVES * 8);
	L.mask = L.leafMeta + VGX_POOL_LEAVES;
	L.flags = L.mask + 64;
	return L;
}

// Subdivides the cubics of the lanes in rootMask (root control points P1..P4 and tessTol of my lane's cubic). Fills the
// leaf list, mask[] and flags[] (both zeroed here). Returns the number of listed leaves.
__device__ __forceinline__ int pool_walk(const PoolLds& L, int lane, uint64_t rootMask, v2f P1r, v2f P2r, v2f P3r, v2f P4r, float tolr)
{
	int top = __popcll(rootMask), leaves = 0;
	L.mask[lane] = 0; L.flags[lane] = 0;
	if ((rootMask >> lane) & 1ull) {
		const int ti = __popcll(rootMask & lanemask_lt(lane));
		L.task[2 * ti] = make_float4(P1r.x, P1r.y, P2r.x, P2r.y);
		L.task[2 * ti + 1] = make_float4(P3r.x, P3r.y, P4r.x, P4r.y);
		PoolTaskTail t; t.meta = (uint32_t)lane; t.tol = tolr;
		L.tail[ti] = t;
	}
	__syncthreads(); // one-wave workgroup: LDS wait only
	while (top > 0) {
		const int n = top < VGX_WAVE ? top : VGX_WAVE;
		const bool active = lane < n;
		const int ti = top - 1 - lane;
		float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a;
		PoolTaskTail tt; tt.meta = 0; tt.tol = 0.0f;
		if (active) { a = L.task[2 * ti]; b = L.task[2 * ti + 1]; tt = L.tail[ti]; }
		const uint32_t cl = tt.meta & 63u, depth = (tt.meta >> 6) & 7u, path = tt.meta >> 9;
		v2f P1, P2, P3, P4;
		P1.x = a.x; P1.y = a.y; P2.x = a.z; P2.y = a.w; P3.x = b.x; P3.y = b.y; P4.x = b.z; P4.y = b.w;
		// flatness (path.cpp:105-116) and de Casteljau halves (:118-129), as in build_flatten_hot
		const v2f d = P4 - P1;
		const v2f a2 = P2 - P4, a3 = P3 - P4;
		const v2f dsw = d.yx;
		const v2f m2 = a2 * dsw, m3 = a3 * dsw;
		const float d2 = __builtin_fabsf(m2.x - m2.y), d3 = __builtin_fabsf(m3.x - m3.y);
		const float d23 = d2 + d3;
		const v2f dd = d * d;
		const bool flat = d23 * d23 <= tt.tol * (dd.x + dd.y);
		const v2f P12 = (P1 + P2) * 0.5f, P23 = (P2 + P3) * 0.5f, P34 = (P3 + P4) * 0.5f;
		const v2f P123 = (P12 + P23) * 0.5f, P234 = (P23 + P34) * 0.5f;
		const v2f P1234 = (P123 + P234) * 0.5f;
		const bool isLeaf = active && flat;
		bool split = active && !flat && depth < VGX_POOL_MAXD;
		const int newBase = top - n;
		if (split && newBase + 2 * (int)__popcll(wave_ballot(split) & lanemask_le(lane)) > VGX_POOL_CAP) { split = false; } // no room
		const uint64_t splitMask = wave_ballot(split);
		const uint64_t leafMask = wave_ballot(isLeaf);
		const int li = leaves + (int)__popcll(leafMask & lanemask_lt(lane));
		const bool listed = isLeaf && li < VGX_POOL_LEAVES;
		const uint32_t start = path << (VGX_POOL_MAXD - depth);
		if (listed) {
			L.leaf[li] = make_float2(P4.x, P4.y);
			L.leafMeta[li] = cl | (start << 6);
			atomicOr(&L.mask[cl], 1u << start);
			const v2f e = P1 - P4;
			const v2f ee = e * e;
			if (ee.x + ee.y < VGM_EPSILON) { atomicOr(&L.flags[cl], VGX_POOL_F_SLOW); }
		}
		if (active && ((!flat && !split) || (isLeaf && !listed))) { atomicOr(&L.flags[cl], VGX_POOL_F_DEEP); } // to the per-lane walk
		if (split) {
			const int pp = newBase + 2 * (int)__popcll(splitMask & lanemask_lt(lane));
			const uint32_t cm = cl | ((depth + 1) << 6);
			L.task[2 * pp] = make_float4(P1.x, P1.y, P12.x, P12.y);
			L.task[2 * pp + 1] = make_float4(P123.x, P123.y, P1234.x, P1234.y);
			L.task[2 * pp + 2] = make_float4(P1234.x, P1234.y, P234.x, P234.y);
			L.task[2 * pp + 3] = make_float4(P34.x, P34.y, P4.x, P4.y);
			PoolTaskTail t0, t1;
			t0.meta = cm | ((path << 1) << 9); t0.tol = tt.tol;
			t1.meta = cm | (((path << 1) | 1u) << 9); t1.tol = tt.tol;
			L.tail[pp] = t0;
			L.tail[pp + 1] = t1;
		}
		top = newBase + 2 * (int)__popcll(splitMask);
		leaves += (int)__popcll(leafMask);
		__syncthreads();
	}
	return leaves < VGX_POOL_LEAVES ? leaves : VGX_POOL_LEAVES;
}

// Writes the listed leaves to their final places. Per OWNER lane (in registers): mask / pooled (a cubic that stayed on the
// pooled path) / base (first vertex of the command, any consistent origin) / limit (vertices [0, limit) are written)
// / mtx (the draw's transform). out.put(index, transformed point).
template<class OUT>
__device__ __forceinline__ void pool_place(const PoolLds& L, int lane, int leaves, uint32_t mask, bool pooled, uint32_t base, uint32_t limit, const float* mtx, OUT& out)
{
	const float m0 = mtx[0], m1 = mtx[1], m2 = mtx[2], m3 = mtx[3], m4 = mtx[4], m5 = mtx[5];
	const uint32_t lim = pooled ? limit : 0u;
	for (int l0 = 0; l0 < leaves; l0 += VGX_WAVE) {
		const int li = l0 + lane;
		const bool valid = li < leaves;
		float2 q = make_float2(0.0f, 0.0f);
		uint32_t lm = 0;
		if (valid) { q = L.leaf[li]; lm = L.leafMeta[li]; }
		const int cl = (int)(lm & 63u);
		const uint32_t start = lm >> 6;
		const uint32_t om = (uint32_t)__shfl((int)mask, cl), ob = (uint32_t)__shfl((int)base, cl), ol = (uint32_t)__shfl((int)lim, cl);
		const float t0 = __shfl(m0, cl), t1 = __shfl(m1, cl), t2 = __shfl(m2, cl), t3 = __shfl(m3, cl), t4 = __shfl(m4, cl), t5 = __shfl(m5, cl);
		const uint32_t rank = (uint32_t)__popc(om & ((1u << start) - 1u));
		if (valid && rank < ol) {
			// transformPos2D, vg_util.h:24-28: (m0 * x + m2 * y) + m4
			out.put(ob + rank, v2(t0 * q.x + t2 * q.y + t4, t1 * q.x + t3 * q.y + t5));
		}
	}
}

struct PoolOutGlobal // destination: polyline heap
{
	float2* p;
	__device__ __forceinline__ void put(uint32_t i, V2 v) const { p[i] = make_float2(v.x, v.y); }
};

} // namespace

#endif
