// vgx_fused.hip -- vgx_tessellate as ONE pass on gfx950: flatten -> transformPath -> strokerXXX with the polyline of a
// whole group of draws staged in LDS (replaces, for one batch, vg::pathXXX + transformPath + vg::strokerXXX, reference
// src/path.cpp, src/vg.cpp:4957-4975, src/stroker.cpp).
//
// Work decomposition
//   The command-instance stream is cut into SEGMENTS (whole draws whose first command falls into one bucket of
//   seg_items commands, at most VGX_FUSED_T draws / sub-path records / meshes -- the bucket size is picked by
//   vgx_tessellate_count so that this holds). A one-wave workgroup takes segments in order from a ticket counter and
//   does, for each:
//     F. flatten: one lane = one path command (the walk of vgx_walk.h, leaves in per-lane LDS slots, ballots + prefix
//        scans segmented by draw / sub-path exactly as k_flatten_build); the transformed polyline goes to the wave's
//        LDS window (VGX_FUSED_P vertices) -- never to HBM;
//     M. mesh records of the segment (one lane per sub-path record), Round-join meshes sized from the window, prefix
//        scans over the segment's meshes -> the segment's totals {meshes, vertices, indices};
//     L. decoupled look-back over the predecessors' totals (8-byte self-validating granules, agent-scope relaxed
//        atomics, no fences) -> where this segment's meshes start in the caller's streams;
//     E. emit: mesh table, then one lane = one polyline vertex of one mesh (vgx_elem.h) reading the window with
//        ds_read and writing positions / colours / indices straight to their final place.
//   The output is byte-identical to the multi-kernel pipeline: segments are in draw order, a draw's meshes are fill
//   meshes by sub-path then stroke meshes (vg.cpp:3099-3131, 3448-3485).
//
// Off the fast path, per segment (wave-uniform decisions):
//   - a draw that needs the exact sequential builder (ARC / ARC_TO / closed shapes, or an epsilon de-dup hit / dropped
//     subdivision piece, see vgx_flatten.hip): the segment is rebuilt with one lane per draw running PathSim;
//   - a polyline that does not fit the LDS window: flattened again into a block of the polyline heap (sized by
//     vgx_tessellate_count), and the element code reads HBM instead of LDS (second instantiation of the emit code);
//   - more draws / sub-path records / meshes than the tables hold: VGX_E_NOSPACE ("run vgx_tessellate_count again").
//   Every taken segment publishes its totals whatever happens, so no wave ever waits for a segment that gave up; the
//   wait itself is bounded (VGX_E_INTERNAL after ~2 s instead of a hung device).
#include "vgx_internal.h"
#include "vgx_wave.h"
#include "vgx_walk.h"
#include "vgx_elem.h"
#include "vgx_pathsim.h"

namespace {

#ifdef VGX_FUSED_PROFILE
#define PROF_T(var) const uint64_t var = wall_clock64()
#define PROF_ADD(A, slot, t0, t1) do { if (threadIdx.x == 0) { atomicAdd(&(A).totals->prof[slot], (unsigned long long)((t1) - (t0))); } } while (0)
#define PROF_INC(A, slot, n) do { if (threadIdx.x == 0) { atomicAdd(&(A).totals->prof[slot], (unsigned long long)(n)); } } while (0)
#else
#define PROF_T(var) do {} while (0)
#define PROF_ADD(A, slot, t0, t1) do {} while (0)
#define PROF_INC(A, slot, n) do {} while (0)
#endif

#define FUSED_P VGX_FUSED_P
#define FUSED_T VGX_FUSED_T
#ifndef FUSED_LV
#define FUSED_LV 3
#endif
/* FUSED_LV:  LDS levels of the walk's pending stack (deeper cubics are redone with the private-memory stack) */

struct FDraw { uint32_t num_fill, num_stroke, mesh_base, serial; };

struct __attribute__((aligned(16))) FMesh // one mesh of the segment, 64 bytes, in LDS during M / E
{
	uint32_t polyFirst, N, kind, draw;       // kind: VgxMeshDesc::kind encoding (VGX_MD_*)
	float hsw, hswAA, fringe; uint32_t color;// fills: hsw = aa (fringe / 2 * orientation sign)
	uint32_t vOff, iOff, nv, ni;             // offsets inside the segment's output
	uint32_t sub, pad0, pad1, pad2;
};

#define FUSED_STACK_F2 (FUSED_LV * 3 * VGX_WAVE)
#define FUSED_LEAF_F2 (VGX_LEAF_SLOTS * VGX_WAVE)
#define FUSED_UNION_BYTES ((FUSED_STACK_F2 + FUSED_LEAF_F2) * 8 > FUSED_T * 64 ? (FUSED_STACK_F2 + FUSED_LEAF_F2) * 8 : FUSED_T * 64)

struct FusedPrivStack
{
	float s[VGX_CUBIC_MAX_PENDING * 6];
	__device__ __forceinline__ void push(int level, float ax, float ay, float bx, float by, float cx, float cy)
	{
		float* p = s + level * 6;
		p[0] = ax; p[1] = ay; p[2] = bx; p[3] = by; p[4] = cx; p[5] = cy;
	}
	__device__ __forceinline__ void pop(int level, float& ax, float& ay, float& bx, float& by, float& cx, float& cy)
	{
		const float* p = s + level * 6;
		ax = p[0]; ay = p[1]; bx = p[2]; by = p[3]; cx = p[4]; cy = p[5];
	}
};

__device__ __forceinline__ void fused_set_status(VgxTotals* t, uint32_t err) { atomicCAS(&t->status, (uint32_t)VGX_OK, err); }
// first failure wins; reason / segment / aux are diagnostics only (vgx_get_failure_info)
__device__ __forceinline__ void fused_fail(VgxTotals* t, uint32_t err, uint32_t reason, uint64_t seg, uint32_t aux)
{
	if (atomicCAS(&t->fail_reason, 0u, reason) == 0u) { t->fail_segment = seg; t->fail_aux = aux; }
	atomicCAS(&t->status, (uint32_t)VGX_OK, err);
}

// ---- look-back: two levels, self-validating 8-byte words, no fences ---------------------------------------------
// Segment s publishes, right after its meshes are sized,
//   agg[s]            = its totals {meshes 8 bits, vertices 26, indices 29} | valid, one agent-scope store, and
//   blk0/blk1[s / 64] += {1 << 57 | vertices (32 bits) | meshes << 32 (16 bits)} / {1 << 57 | indices (40 bits)}: one
//                       atomic add per word; a block's word is complete when its count field equals the block's size.
// Its exclusive prefix = sum of agg over the earlier segments of its own block (one 512-byte read, one lane each)
//                      + prefix at the end of the previous block: the nearest earlier block whose prefix is known
//                        (pre0/pre1[b]: {vertices 40 | meshes low 23} / {indices 42 | meshes high 21} | valid) plus the
//                        complete sums of the blocks in between, 64 blocks per round.
// Every wave that learns the prefix at the end of block b - 1 publishes it, so a stalled range of thousands of
// segments drains in two or three hops once the slow segment arrives, instead of one hop per 64 segments. All words
// are read with agent-scope loads (L2 / fabric, never a stale L1 line) and validate themselves: no fence, no L2
// write-back of the streaming output. Waiting waves back off (s_sleep) so that polling does not eat the memory pipeline.
__device__ __forceinline__ void granule_store(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t granule_load(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#define FUSED_AGG_MAX_V ((1u << 26) - 1)
#define FUSED_AGG_MAX_I ((1u << 29) - 1)
#define FUSED_BLK 64

__device__ __forceinline__ uint64_t agg_pack(uint32_t m, uint32_t v, uint32_t i)
{
	return 1ull | ((uint64_t)(m & 0xFFu) << 1) | ((uint64_t)(v & FUSED_AGG_MAX_V) << 9) | ((uint64_t)(i & FUSED_AGG_MAX_I) << 35);
}
__device__ __forceinline__ uint64_t pre0_pack(uint64_t m, uint64_t v) { return 1ull | ((v & ((1ull << 40) - 1)) << 1) | ((m & ((1ull << 23) - 1)) << 41); }
__device__ __forceinline__ uint64_t pre1_pack(uint64_t m, uint64_t i) { return 1ull | ((i & ((1ull << 42) - 1)) << 1) | (((m >> 23) & ((1ull << 21) - 1)) << 43); }

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { v += __shfl_xor((unsigned long long)v, d); }
	return v;
}

struct LookState // device arrays of one call (zeroed before the launch)
{
	uint64_t* agg;   // [segments]
	uint64_t* blk0;  // [blocks]
	uint64_t* blk1;
	uint64_t* pre0;  // [blocks] prefix at the END of block b
	uint64_t* pre1;
};

__device__ __forceinline__ LookState look_state(const VgxFusedArgs& A)
{
	LookState L;
	const uint64_t nb = A.seg_cap / FUSED_BLK + 2;
	L.agg = A.seg_state;
	L.blk0 = A.seg_state + (A.seg_cap + 2);
	L.blk1 = L.blk0 + nb;
	L.pre0 = L.blk1 + nb;
	L.pre1 = L.pre0 + nb;
	return L;
}

__device__ __forceinline__ void fused_publish(const LookState& L, uint64_t seg, int lane, uint32_t m, uint32_t v, uint32_t i)
{
	if (lane == 0) {
		granule_store(L.agg + seg, agg_pack(m, v, i));
		const uint64_t b = seg / FUSED_BLK;
		atomicAdd((unsigned long long*)(L.blk0 + b), (1ull << 57) | (uint64_t)v | ((uint64_t)m << 32));
		atomicAdd((unsigned long long*)(L.blk1 + b), (1ull << 57) | (uint64_t)i);
	}
}

// Exclusive prefix {meshes, vertices, indices} of segment `seg`. Returns false on timeout (a predecessor never
// published: cannot happen unless a wave died).
__device__ __forceinline__ bool fused_lookback(const LookState& L, uint64_t seg, uint64_t numSegments, int lane, uint64_t* baseM, uint64_t* baseV, uint64_t* baseI)
{
	const uint64_t t0 = wall_clock64();
	uint32_t spins = 0;
	uint64_t sumM = 0, sumV = 0, sumI = 0;
	// level 1: the earlier segments of my own block
	const uint64_t b = seg / FUSED_BLK;
	const int k = (int)(seg - b * FUSED_BLK);
	if (k > 0) {
		for (;;) {
			uint64_t a = 1;
			if (lane < k) { a = granule_load(L.agg + (seg - 1 - (uint64_t)lane)); }
			if (wave_ballot((a & 1ull) == 0) == 0) {
				const bool take = lane < k;
				sumM = wave_sum_u64(take ? ((a >> 1) & 0xFFu) : 0);
				sumV = wave_sum_u64(take ? ((a >> 9) & FUSED_AGG_MAX_V) : 0);
				sumI = wave_sum_u64(take ? (a >> 35) : 0);
				break;
			}
			__builtin_amdgcn_s_sleep(32);
			if ((++spins & 63u) == 0 && wall_clock64() - t0 > 200000000ull) { return false; } // 2 s at 100 MHz
		}
	}
	// level 2: prefix at the end of block b - 1
	if (b > 0) {
		uint64_t bm = 0, bv = 0, bi = 0;
		int64_t pos = (int64_t)b - 1; // lane l looks at block pos - l
		for (;;) {
			const int64_t idx = pos - lane;
			bool hasP = idx < 0, full = false;
			uint64_t m = 0, v = 0, i = 0;
			if (idx >= 0) {
				const uint64_t p0 = granule_load(L.pre0 + idx), p1 = granule_load(L.pre1 + idx);
				if ((p0 & 1ull) && (p1 & 1ull)) {
					hasP = true;
					v = (p0 >> 1) & ((1ull << 40) - 1);
					i = (p1 >> 1) & ((1ull << 42) - 1);
					m = (p0 >> 41) | ((p1 >> 43) << 23);
				} else {
					const uint64_t w0 = granule_load(L.blk0 + idx), w1 = granule_load(L.blk1 + idx);
					const uint64_t segsInBlock = ((uint64_t)idx + 1) * FUSED_BLK <= numSegments ? (uint64_t)FUSED_BLK : numSegments - (uint64_t)idx * FUSED_BLK;
					if ((w0 >> 57) == segsInBlock && (w1 >> 57) == segsInBlock) {
						full = true;
						v = w0 & 0xFFFFFFFFull; m = (w0 >> 32) & 0xFFFFull; i = w1 & ((1ull << 40) - 1);
					}
				}
			}
			const uint64_t mP = wave_ballot(hasP), mR = wave_ballot(hasP || full);
			const int firstP = mP ? (int)__builtin_ctzll(mP) : VGX_WAVE;
			const uint64_t need = (firstP >= VGX_WAVE) ? ~0ull : lanemask_le(firstP);
			if ((mR & need) != need) { // a block in front of the nearest known prefix is not complete yet
				__builtin_amdgcn_s_sleep(64);
				if ((++spins & 63u) == 0 && wall_clock64() - t0 > 200000000ull) { return false; }
				continue;
			}
			const bool take = lane <= firstP;
			bm += wave_sum_u64(take ? m : 0);
			bv += wave_sum_u64(take ? v : 0);
			bi += wave_sum_u64(take ? i : 0);
			if (firstP < VGX_WAVE) { break; }
			pos -= VGX_WAVE;
		}
		if (lane == 0) { // what I just learnt helps everybody behind me
			granule_store(L.pre0 + (b - 1), pre0_pack(bm, bv));
			granule_store(L.pre1 + (b - 1), pre1_pack(bm, bi));
		}
		sumM += bm; sumV += bv; sumI += bi;
	}
	*baseM = sumM; *baseV = sumV; *baseI = sumI;
	return true;
}

// ---- F: flatten one segment into the polyline window (or a heap block) ------------------------------------------
struct FusedWindow { uint64_t prefix; uint32_t pc0; uint32_t serial; };

struct FlatResult
{
	int cursor;          // window positions used (holes from popped vertices included): the heap block a re-run needs
	int polyTrue;        // pathGetNumVertices summed over the segment's draws
	uint32_t subRecs;    // sub-path records written to s_sub
	uint32_t numSubs;    // pathGetNumSubPaths summed over the draws
	bool anySerial, overflowP, overflowT;
};

__device__ __forceinline__ FlatResult fused_flatten(const VgxFusedArgs& A, uint64_t d0, uint64_t C0, uint64_t C1, const FusedWindow& W, int lane,
	vgx_f2* s_poly, float2* s_stack, float2* s_leaf, VgxSegSub* s_sub, FDraw* s_draw, bool toHeap, float2* heapDst, uint32_t heapCap)
{
	const VgxPathSetDev& ps = A.ps;
	LdsStackT<FUSED_LV> stack;
	stack.base = &s_stack[lane];
	FlatResult R;
	R.cursor = 0; R.polyTrue = 0; R.subRecs = 0; R.numSubs = 0; R.anySerial = false; R.overflowP = false; R.overflowT = false;
	int cur = 0;
	int carryDrawVerts = 0, carrySpVerts = 0, carrySubs = 0, carryFill = 0, carryStroke = 0, carrySlow = 0;
	const uint32_t cap = toHeap ? heapCap : (uint32_t)FUSED_P;

	for (uint64_t chunk = C0; chunk < C1; chunk += VGX_WAVE) {
		const uint64_t ci = chunk + lane;
		const bool valid = ci < C1;
		PROF_T(tc0);
		// ---- decode: owner draw from the lane-resident window (all draws of the segment are in it) ----------------
		const uint32_t wrel = window_rel(W.prefix, chunk);
		const int ownerOfs = window_owner_rel(wrel, valid ? (uint32_t)lane : 0u);
		const uint32_t orel = (uint32_t)__shfl((int)wrel, ownerOfs);
		const int firstOwner = __popcll(wave_ballot(W.prefix <= chunk)) - 1;
		const uint64_t ownerBase = orel > 0 ? chunk + orel : wave_bcast_u64(W.prefix, firstOwner < 0 ? 0 : firstOwner);
		uint32_t pc0 = (uint32_t)__shfl((int)(W.pc0 | (W.serial << 31)), ownerOfs);
		const bool serialDraw = (pc0 >> 31) != 0;
		pc0 &= 0x7FFFFFFFu;
		uint32_t type = VGX_CMD_CLOSE, cflags = 0, na = 0;
		bool drawHead = false, drawLast = false;
		float scale = 1.0f, tol = 0.25f;
		uint32_t fillFlags = 0, strokeFlags = 0;
		const vgx_draw* dr = A.draws;
		VgxCmdRec rec;
		rec.type = VGX_CMD_CLOSE; rec.flags = 0; rec.na = 0; rec.arg_off = 0; rec.start[0] = 0.0f; rec.start[1] = 0.0f;
		for (int i = 0; i < 8; ++i) { rec.a[i] = 0.0f; }
		const uint32_t dlocal = (uint32_t)ownerOfs;
		if (valid) {
			dr = A.draws + (d0 + (uint64_t)ownerOfs);
			const uint32_t k = (uint32_t)(ci - ownerBase);
			rec = ps.cmdrec[pc0 + k];
			type = rec.type; cflags = rec.flags; na = rec.na;
			drawHead = (k == 0);
			drawLast = (cflags & VGX_CF_LAST_IN_PATH) != 0;
			scale = dr->scale; tol = dr->tess_tol;
			fillFlags = dr->fill_flags; strokeFlags = dr->stroke_flags;
		}
		const float* a = rec.a;
		const float* pa = ps.args + rec.arg_off;
		const float* mtx = dr->mtx;
		const V2 start = v2(rec.start[0], rec.start[1]);

#ifdef VGX_FUSED_PROFILE
		{ float sink_ = rec.a[0] + scale + (float)fillFlags; asm volatile("" :: "v"(sink_)); } // the decode loads have landed
#endif
		PROF_T(tc1);
		PROF_ADD(A, 8, tc0, tc1);
		// ---- subdivide ONCE: count, detect degenerate cases, keep the first leaves in LDS ------------------------
		int cnt = 0;
		bool slow = false, exists = false, closedHere = false;
		float c1x = a[0], c1y = a[1], c2x = a[2], c2y = a[3], ex = a[4], ey = a[5];
		float2* over = (float2*)A.leaf_overflow + (size_t)blockIdx.x * VGX_BUILD_OVERFLOW * VGX_WAVE + lane;
		if (valid && !serialDraw) {
			switch (type) {
			case VGX_CMD_MOVE_TO: cnt = 1; exists = true; break;
			case VGX_CMD_LINE_TO: cnt = 1; slow = v2near(start, v2(a[0], a[1])); break;
			case VGX_CMD_CUBIC_TO:
			case VGX_CMD_QUAD_TO: {
				if (type == VGX_CMD_QUAD_TO) {
					ex = a[2]; ey = a[3];
					vgx_quad_to_cubic(start.x, start.y, a[0], a[1], ex, ey, &c1x, &c1y, &c2x, &c2y);
				}
				const float tessTol = tol / (scale * scale);
				uint32_t nLeaves = 0;
				v2f q1, q2, q3, q4;
				q1.x = start.x; q1.y = start.y; q2.x = c1x; q2.y = c1y; q3.x = c2x; q3.y = c2y; q4.x = ex; q4.y = ey;
				if (!build_flatten_hot<FUSED_LV>(q1, q2, q3, q4, tessTol, &s_stack[lane], &s_leaf[lane], over, &nLeaves, &slow)) {
					BuildCubicSink sink; // nests deeper than the LDS levels: full-depth walk from the root
					sink.prev = start; sink.n = 0; sink.slow = false; sink.slots = &s_leaf[lane]; sink.over = over;
					vgx_flatten_cubic(start.x, start.y, c1x, c1y, c2x, c2y, ex, ey, tessTol, stack, sink);
					nLeaves = sink.n;
					slow = sink.slow;
				}
				cnt = (int)nLeaves;
			} break;
			case VGX_CMD_POLYLINE: {
				const uint32_t npts = na >> 1;
				cnt = (int)npts - ((npts > 0 && v2near(start, v2(pa[0], pa[1]))) ? 1 : 0);
				slow = cnt == 0;
			} break;
			default: break;
			}
		}
		const int rawCnt = cnt;
		PROF_T(tc2);
		PROF_ADD(A, 9, tc1, tc2);

		// ---- segmented bookkeeping (as k_flatten_build) -----------------------------------------------------------
		const uint64_t drawHeads = wave_ballot(valid && drawHead);
		const uint64_t subHeads = wave_ballot(valid && (cflags & VGX_CF_STARTS_SUB));
		const int dh = seg_head(drawHeads, lane);
		const int sh = seg_head(subHeads, lane);
		{
			const int incl1 = wave_incl_scan(cnt, lane);
			const int spBefore1 = seg_rel(incl1 - cnt, sh, carrySpVerts);
			if (valid && !serialDraw && type == VGX_CMD_CLOSE && spBefore1 > 2) { // pathClose, path.cpp:707-726
				closedHere = true;
				if (v2near(start, v2(rec.a[6], rec.a[7]))) { cnt = -1; } // pop: the previous vertex is removed
			}
		}
		const int incl = wave_incl_scan(cnt, lane);
		const int excl = incl - cnt;
		const int inDrawBefore = seg_rel(excl, dh, carryDrawVerts);
		const int spBefore = seg_rel(excl, sh, carrySpVerts);
		const int spTotal = spBefore + cnt;
		const uint64_t existMask = wave_ballot(valid && exists);
		const uint64_t mine = seg_mask_upto(dh, lane);
		const int subsIncl = __popcll(existMask & mine) + (dh < 0 ? carrySubs : 0);
		const bool lastInSub = valid && !serialDraw && (cflags & VGX_CF_LAST_IN_SUB);
		const bool hasFill = lastInSub && (fillFlags & VGX_FILL_ENABLE) && spTotal >= 3;
		const bool hasStroke = lastInSub && (strokeFlags & VGX_STROKE_ENABLE) && spTotal >= 2;
		const uint64_t fillMask = wave_ballot(hasFill);
		const uint64_t strokeMask = wave_ballot(hasStroke);
		const int fillIncl = __popcll(fillMask & mine) + (dh < 0 ? carryFill : 0);
		const int strokeIncl = __popcll(strokeMask & mine) + (dh < 0 ? carryStroke : 0);
		const uint64_t slowMask = wave_ballot(valid && slow);
		const bool slowDraw = ((slowMask & mine) != 0) || (dh < 0 && carrySlow);

		const int nvalid = (int)((C1 - chunk) < (uint64_t)VGX_WAVE ? (C1 - chunk) : (uint64_t)VGX_WAVE);
		const int L = nvalid - 1;
		const int chunkTotal = wave_bcast(incl, L);
		const int pops = __popcll(wave_ballot(cnt < 0));
		if ((uint32_t)(cur + chunkTotal + pops) > cap) { R.overflowP = true; } // from here on the window is only counted
		const bool store = !R.overflowP;
		PROF_T(tc3);
		PROF_ADD(A, 10, tc2, tc3);
		{
			const int g = cur + excl; // window index of my first vertex
			if (valid && !serialDraw && store) {
				// my last vertex is the one pathClose removes (same decision the CLOSE lane takes)
				uint32_t limit = (uint32_t)(rawCnt < 0 ? 0 : rawCnt);
				if ((cflags & VGX_CF_NEXT_IS_CLOSE) && limit > 0 && spTotal > 2) {
					const V2 endp = (type == VGX_CMD_POLYLINE) ? v2(pa[na - 2], pa[na - 1]) : (type == VGX_CMD_CUBIC_TO ? v2(a[4], a[5]) : (type == VGX_CMD_QUAD_TO ? v2(a[2], a[3]) : v2(a[0], a[1])));
					if (v2near(endp, v2(rec.a[6], rec.a[7]))) { --limit; }
				}
				if (type == VGX_CMD_MOVE_TO || type == VGX_CMD_LINE_TO) {
					if (limit > 0) {
						const V2 p = v2xform(v2(a[0], a[1]), mtx);
						if (toHeap) { heapDst[g] = make_float2(p.x, p.y); } else { vgx_f2 t; t.x = p.x; t.y = p.y; s_poly[g] = t; }
					}
				} else if (type == VGX_CMD_CUBIC_TO || type == VGX_CMD_QUAD_TO) {
					if ((uint32_t)rawCnt <= VGX_LEAF_SLOTS + VGX_BUILD_OVERFLOW) {
						const uint32_t nl = limit < VGX_LEAF_SLOTS ? limit : VGX_LEAF_SLOTS;
						for (uint32_t i = 0; i < nl; ++i) {
							const float2 q = s_leaf[i * VGX_WAVE + lane];
							const V2 p = v2xform(v2(q.x, q.y), mtx);
							if (toHeap) { heapDst[g + (int)i] = make_float2(p.x, p.y); } else { vgx_f2 t; t.x = p.x; t.y = p.y; s_poly[g + (int)i] = t; }
						}
						for (uint32_t i = VGX_LEAF_SLOTS; i < limit; ++i) { // same lane wrote these during its subdivision
							const float2 q = over[(i - VGX_LEAF_SLOTS) * VGX_WAVE];
							const V2 p = v2xform(v2(q.x, q.y), mtx);
							if (toHeap) { heapDst[g + (int)i] = make_float2(p.x, p.y); } else { vgx_f2 t; t.x = p.x; t.y = p.y; s_poly[g + (int)i] = t; }
						}
					} else { // more leaves than slots + overflow area: subdivide again, straight to the destination
						FastCubicSink<true, true> sink;
						sink.prev = start; sink.n = 0; sink.slow = false; sink.writeLimit = limit; sink.mtx = mtx;
						sink.out = toHeap ? (float*)(heapDst + g) : (float*)(s_poly + g);
						wave_flatten_cubic(start.x, start.y, c1x, c1y, c2x, c2y, ex, ey, tol / (scale * scale), stack, sink);
					}
				} else if (type == VGX_CMD_POLYLINE) {
					const uint32_t skip = (na >> 1) - (uint32_t)rawCnt;
					for (uint32_t i = 0; i < limit; ++i) {
						const V2 p = v2xform(v2(pa[2 * (i + skip)], pa[2 * (i + skip) + 1]), mtx);
						if (toHeap) { heapDst[g + (int)i] = make_float2(p.x, p.y); } else { vgx_f2 t; t.x = p.x; t.y = p.y; s_poly[g + (int)i] = t; }
					}
				}
			}
			// sub-path records of the sub-paths that produce a mesh, in order
			const uint64_t recMask = fillMask | strokeMask;
			if (hasFill || hasStroke) {
				const uint32_t ri = R.subRecs + (uint32_t)__popcll(recMask & lanemask_lt(lane));
				if (ri < FUSED_T) {
					VgxSegSub sr;
					sr.first = (uint32_t)(g - spBefore);
					sr.info = (uint32_t)spTotal | (closedHere ? 0x80000000u : 0u);
					sr.packed = dlocal | (((uint32_t)(fillIncl - 1) & 0xFFFu) << 8) | (((uint32_t)(strokeIncl - 1) & 0xFFFu) << 20);
					sr.sub_index = (uint32_t)(subsIncl - 1);
					s_sub[ri] = sr;
				}
			}
			R.subRecs += (uint32_t)__popcll(recMask);
			if (valid && drawLast) {
				FDraw fd;
				fd.num_fill = (uint32_t)fillIncl; fd.num_stroke = (uint32_t)strokeIncl; fd.mesh_base = 0;
				fd.serial = (serialDraw || slowDraw) ? 1u : 0u;
				s_draw[dlocal] = fd;
			}
			if (wave_ballot(valid && drawLast && (serialDraw || slowDraw)) != 0) { R.anySerial = true; }
			R.numSubs += (uint32_t)__popcll(existMask);
			cur += chunkTotal > 0 ? chunkTotal : 0;
			R.polyTrue += chunkTotal;
		}

		PROF_T(tc4);
		PROF_ADD(A, 11, tc3, tc4);
		PROF_INC(A, 12, 1);
		// carries into the next chunk
		const int lastIsDrawLast = wave_bcast((int)drawLast, L);
		const int lastIsSubLast = wave_bcast((int)((cflags & VGX_CF_LAST_IN_SUB) != 0), L);
		const int nDraw = wave_bcast(inDrawBefore + cnt, L);
		const int nSp = wave_bcast(spTotal, L);
		const int nSubs = wave_bcast(subsIncl, L);
		const int nFill = wave_bcast(fillIncl, L);
		const int nStroke = wave_bcast(strokeIncl, L);
		const int nSlow = wave_bcast((int)slowDraw, L);
		carryDrawVerts = lastIsDrawLast ? 0 : nDraw;
		carrySubs = lastIsDrawLast ? 0 : nSubs;
		carryFill = lastIsDrawLast ? 0 : nFill;
		carryStroke = lastIsDrawLast ? 0 : nStroke;
		carrySlow = lastIsDrawLast ? 0 : nSlow;
		carrySpVerts = (lastIsDrawLast || lastIsSubLast) ? 0 : nSp;
	}
	R.cursor = cur;
	if (R.subRecs > FUSED_T) { R.overflowT = true; }
	return R;
}

// ---- exact rebuild of a segment with the sequential builder, one lane per draw -----------------------------------
// (paths with ARC / ARC_TO / closed shapes, and draws the lane-parallel walk flagged as degenerate). Writes the same
// things fused_flatten does: the transformed polyline (window or heap block), sub-path records, per-draw mesh counts.
// Results come back through LDS (s_res) and every argument is passed by value: taking the address of the kernel's
// argument block would push it to private memory and turn the kernel's own global accesses into flat ones.
enum { SR_POLY = 0, SR_SUBRECS, SR_NUMSUBS, SR_NUMSERIAL, SR_FLAGS, SR_HEAP_LO, SR_HEAP_HI, SR_COUNT };
enum { SRF_TOHEAP = 1, SRF_FAILED = 2, SRF_OVERFLOW_T = 4 };

__device__ __noinline__ void fused_serial_segment(const vgx_draw* draws, const uint32_t* path_cmd_begin, const uint8_t* cmd_type, const uint32_t* cmd_arg_off, const float* args,
	float* heap, uint64_t heap_cap, VgxTotals* totals, uint64_t d0, uint32_t nd, float* polyWin, VgxSegSub* s_sub, FDraw* s_draw, uint32_t* s_res)
{
	const int lane = threadIdx.x;
	VgxPathSetDev ps;
	ps.cmdrec = nullptr; ps.cmd_type = cmd_type; ps.cmd_flags = nullptr; ps.cmd_arg_off = cmd_arg_off; ps.cmd_sp_start = nullptr; ps.args = args;
	ps.path_cmd_begin = path_cmd_begin; ps.path_flags = nullptr; ps.path_sub_begin = nullptr; ps.sub_last_cmd = nullptr; ps.npaths = 0; ps.ncmd = 0;
	FusedPrivStack st;
	const bool mineValid = (uint32_t)lane < nd;
	const uint64_t d = d0 + (uint64_t)lane;
	const vgx_draw* dr = draws + (mineValid ? d : d0);
	uint32_t pc0 = 0, pc1 = 0;
	if (mineValid) { const uint32_t path = dr->path; pc0 = path_cmd_begin[path]; pc1 = path_cmd_begin[path + 1]; }
	uint32_t nverts = 0, nsubs = 0, nfill = 0, nstroke = 0, nrec = 0;
	const uint32_t wasSerial = mineValid ? s_draw[lane].serial : 0u;
	if (mineValid && pc0 != pc1) {
		PathSim<false, false> sim;
		sim.scale = dr->scale; sim.tol = dr->tess_tol; sim.mtx = dr->mtx; sim.poly = nullptr;
		sim.drawIndex = (uint32_t)d; sim.fillFlags = dr->fill_flags; sim.strokeFlags = dr->stroke_flags; sim.draw = dr;
		sim.polyBase = 0; sim.subs = nullptr; sim.subBase = 0; sim.mdesc = nullptr; sim.mprep = nullptr; sim.mtab = nullptr; sim.meshBase = 0; sim.numFillTotal = 0; sim.limit = 0;
		sim.init();
		sim.segSubs = s_sub; sim.segSubCap = 0; // count only
		sim.run(ps, pc0, pc1, st);
		nverts = sim.nverts; nsubs = sim.nsubs; nfill = sim.nfill; nstroke = sim.nstroke; nrec = sim.segSubN;
	}
	const uint32_t inclV = wave_incl_scan_u32(nverts, lane), inclR = wave_incl_scan_u32(nrec, lane);
	const uint32_t totalV = wave_bcast_u32(inclV, VGX_WAVE - 1), totalR = wave_bcast_u32(inclR, VGX_WAVE - 1);
	const uint32_t numSubs = (uint32_t)wave_sum_u64(nsubs);
	const uint32_t numSerial = (uint32_t)__popcll(wave_ballot(wasSerial != 0));
	uint32_t flags = totalR > FUSED_T ? (uint32_t)SRF_OVERFLOW_T : 0u;
	float* dst = polyWin;
	unsigned long long base = 0;
	if (totalV > FUSED_P) {
		if (lane == 0) { base = atomicAdd(&totals->poly_heap_cursor, (unsigned long long)totalV); }
		base = wave_bcast_u64(base, 0);
		if (base + totalV > heap_cap) { flags |= SRF_FAILED; }
		else { flags |= SRF_TOHEAP; dst = heap + 2 * base; }
	}
	if (lane == 0) {
		s_res[SR_POLY] = totalV; s_res[SR_SUBRECS] = totalR; s_res[SR_NUMSUBS] = numSubs; s_res[SR_NUMSERIAL] = numSerial;
		s_res[SR_FLAGS] = flags; s_res[SR_HEAP_LO] = (uint32_t)base; s_res[SR_HEAP_HI] = (uint32_t)(base >> 32);
	}
	if (flags & (SRF_FAILED | SRF_OVERFLOW_T)) { return; }
	if (mineValid) {
		FDraw fd;
		fd.num_fill = nfill; fd.num_stroke = nstroke; fd.mesh_base = 0; fd.serial = wasSerial;
		s_draw[lane] = fd;
	}
	if (mineValid && pc0 != pc1) {
		PathSim<true, true> sim;
		sim.scale = dr->scale; sim.tol = dr->tess_tol; sim.mtx = dr->mtx; sim.poly = dst;
		sim.drawIndex = (uint32_t)d; sim.fillFlags = dr->fill_flags; sim.strokeFlags = dr->stroke_flags; sim.draw = dr;
		sim.polyBase = inclV - nverts; sim.subs = nullptr; sim.subBase = 0; sim.mdesc = nullptr; sim.mprep = nullptr; sim.mtab = nullptr; sim.meshBase = 0; sim.numFillTotal = nfill;
		sim.limit = nverts;
		sim.init();
		sim.segSubs = s_sub + (inclR - nrec); sim.segSubCap = nrec; sim.segDrawLocal = (uint32_t)lane;
		sim.run(ps, pc0, pc1, st);
	}
}

// ---- M + L + E: meshes of the segment, look-back, emit ------------------------------------------------------------
template<class VS>
__device__ __forceinline__ VS vs_offset(const VS& base, uint32_t first) { VS r; r.p = base.p + first; return r; }

template<class VS>
__device__ __forceinline__ MeshCtxT<VS> mesh_ctx_from(const FMesh& r, const vgx_draw* draws, const VS& polyBase, uint32_t j)
{
	MeshCtxT<VS> mc;
	mc.kind = VGX_MD_KIND(r.kind);
	mc.closed = VGX_MD_CLOSED(r.kind) != 0;
	mc.cap = VGX_MD_CAP(r.kind);
	mc.join = VGX_MD_JOIN(r.kind);
	mc.N = r.N; mc.j = j;
	mc.hsw = r.hsw; mc.hswAA = r.hswAA; mc.fringe = r.fringe;
	mc.dr = draws + r.draw;
	mc.vtx = vs_offset(polyBase, r.polyFirst);
	return mc;
}

struct SegCounts { uint32_t polyVerts, numSubs, subRecs, numSerial; };
// batch totals summed per WAVE and added to the device totals once, when the wave runs out of tickets: one atomic per
// segment on the same cache line (a million per Tiger x10k step) serialises in the L2's atomic unit and holds up every
// other access to that memory channel (measured: the whole kernel ran at 8.5 segments / us whatever the wave count)
struct WaveTotals { uint64_t polyVerts, numSubs, numSerial, elems, fillElems; };

template<class VS>
__device__ __forceinline__ void fused_finish(const VgxFusedArgs& A, uint64_t seg, uint64_t numSegments, uint64_t d0, uint32_t nd, const SegCounts& sc, bool failed, int lane,
	const VS polyBase, const VgxSegSub* s_sub, FDraw* s_draw, FMesh* s_mesh, WaveTotals& wt)
{
	const LookState LS = look_state(A);
	PROF_T(tM0);
	// ---- M: mesh bases per draw, mesh records -------------------------------------------------------------------
	uint32_t totalMeshes = 0;
	if (!failed) {
		const FDraw fd = ((uint32_t)lane < nd) ? s_draw[lane] : FDraw{0, 0, 0, 0};
		const uint32_t nm = fd.num_fill + fd.num_stroke;
		const uint32_t inclM = wave_incl_scan_u32(nm, lane);
		totalMeshes = wave_bcast_u32(inclM, VGX_WAVE - 1);
		if ((uint32_t)lane < nd) { s_draw[lane].mesh_base = inclM - nm; }
		if (totalMeshes > FUSED_T) { failed = true; fused_fail(A.totals, VGX_E_NOSPACE, VGX_FAIL_SEG_MESHES, seg, totalMeshes); }
	}
	if (!failed) {
		__syncthreads(); // one-wave workgroup: an LDS wait, no barrier instruction (s_draw written above is read by other lanes below)
		if ((uint32_t)lane < sc.subRecs) {
			const VgxSegSub sr = s_sub[lane];
			const uint32_t dl = sr.packed & 0xFFu;
			const FDraw fd = s_draw[dl];
			const uint64_t d = d0 + dl;
			const vgx_draw* dr = A.draws + d;
			const uint32_t n = sr.info & 0x7FFFFFFFu;
			const bool closed = (sr.info >> 31) != 0;
			const uint32_t fillFlags = dr->fill_flags, strokeFlags = dr->stroke_flags;
			if ((fillFlags & VGX_FILL_ENABLE) && n >= 3) {
				FMesh m;
				m.polyFirst = sr.first; m.N = n; m.draw = (uint32_t)d; m.sub = sr.sub_index;
				const uint32_t kind = (fillFlags & VGX_FILL_AA) ? VGX_MESH_FILL_AA : VGX_MESH_FILL;
				m.kind = kind | (closed ? 0x100u : 0u);
				m.hsw = 0.0f; m.hswAA = 0.0f; m.fringe = dr->fringe; m.color = dr->fill_color;
				if (kind == VGX_MESH_FILL_AA) { // orientation from the first triangle only (stroker.cpp:721-723)
					const VS v = vs_offset(polyBase, sr.first);
					const V2 q0 = v.ld(0), q1 = v.ld(1), q2 = v.ld(2);
					m.hsw = dr->fringe * 0.5f * vgm_sign(v2cross(v2sub(q1, q0), v2sub(q2, q0)));
				}
				vgx_mesh_closed_form(kind, closed, 0, 0, n, 2, &m.nv, &m.ni);
				m.vOff = 0; m.iOff = 0; m.pad0 = 0; m.pad1 = 0; m.pad2 = 0;
				s_mesh[fd.mesh_base + ((sr.packed >> 8) & 0xFFFu)] = m;
			}
			if ((strokeFlags & VGX_STROKE_ENABLE) && n >= 2) {
				FMesh m;
				m.polyFirst = sr.first; m.N = n; m.draw = (uint32_t)d; m.sub = sr.sub_index;
				const uint32_t kind = !(strokeFlags & VGX_STROKE_AA) ? VGX_MESH_STROKE : ((strokeFlags & VGX_STROKE_THIN) ? VGX_MESH_STROKE_AA_THIN : VGX_MESH_STROKE_AA);
				const VgxStrokeParams sp = vgx_stroke_params(kind, closed, strokeFlags, dr->stroke_width, dr->fringe, dr->scale, dr->tess_tol);
				m.kind = kind | (closed ? 0x100u : 0u) | (sp.cap << 9) | (sp.join << 11);
				m.hsw = sp.hsw; m.hswAA = sp.hswAA; m.fringe = dr->fringe; m.color = dr->stroke_color;
				const uint32_t H = (!closed && sp.cap == VGX_CAP_ROUND) ? vgx_half_circle_points(vgx_step_angle(dr->scale, sp.hsw, dr->tess_tol)) : 2u;
				if (!vgx_mesh_closed_form(kind, closed, sp.cap, sp.join, n, H, &m.nv, &m.ni)) { m.nv = VGX_MESH_NEEDS_COUNT; m.ni = 0; }
				m.vOff = 0; m.iOff = 0; m.pad0 = 0; m.pad1 = 0; m.pad2 = 0;
				s_mesh[fd.mesh_base + fd.num_fill + (sr.packed >> 20)] = m;
			}
		}
		__syncthreads();
	}
	// Round joins: the only sizes that depend on the geometry (one mesh at a time, lanes stride over its elements)
	uint32_t nv = 0, ni = 0, N = 0, kind = VGX_MESH_FILL;
	if (!failed) {
		uint64_t needCount = wave_ballot((uint32_t)lane < totalMeshes && s_mesh[lane].nv == VGX_MESH_NEEDS_COUNT);
		while (needCount) {
			const int k = (int)__builtin_ctzll(needCount);
			needCount &= needCount - 1;
			const FMesh r = s_mesh[k];
			uint32_t sv, si;
			round_mesh_size(mesh_ctx_from(r, A.draws, polyBase, 0), lane, &sv, &si);
			if (lane == 0) { s_mesh[k].nv = sv; s_mesh[k].ni = si; }
		}
		__syncthreads();
		if ((uint32_t)lane < totalMeshes) { const FMesh r = s_mesh[lane]; nv = r.nv; ni = r.ni; N = r.N; kind = VGX_MD_KIND(r.kind); }
	}
	const bool isFill = kind < VGX_MESH_STROKE;
	const bool tooLarge = wave_ballot(nv > 65536u) != 0; // uint16 indices (vg.cpp:734)
	if (tooLarge) { fused_fail(A.totals, VGX_E_MESH_TOO_LARGE, VGX_FAIL_MESH_TOO_LARGE, seg, 0); failed = true; }
	const uint32_t inclV = wave_incl_scan_u32(failed ? 0u : nv, lane), inclI = wave_incl_scan_u32(failed ? 0u : ni, lane);
	const uint32_t inclF = wave_incl_scan_u32((!failed && isFill) ? N : 0u, lane), inclS = wave_incl_scan_u32((!failed && !isFill) ? N : 0u, lane);
	uint32_t totV = wave_bcast_u32(inclV, VGX_WAVE - 1), totI = wave_bcast_u32(inclI, VGX_WAVE - 1);
	const uint32_t totF = wave_bcast_u32(inclF, VGX_WAVE - 1), totS = wave_bcast_u32(inclS, VGX_WAVE - 1);
	if (failed) { totalMeshes = 0; }
	if (totV > FUSED_AGG_MAX_V || totI > FUSED_AGG_MAX_I) { fused_fail(A.totals, VGX_E_RANGE, VGX_FAIL_AGG_RANGE, seg, totV); failed = true; totV = 0; totI = 0; totalMeshes = 0; }
	const uint32_t vOff = inclV - nv, iOff = inclI - ni;
	// element prefix per mesh lane (exclusive); lanes past the last mesh hold the total, so they never own an element
	const uint32_t preF = ((uint32_t)lane < totalMeshes) ? inclF - (isFill ? N : 0u) : totF;
	const uint32_t preS = ((uint32_t)lane < totalMeshes) ? inclS - (isFill ? 0u : N) : totS;

	// ---- L: publish the aggregate, look back, publish the inclusive prefix -----------------------------------------
	PROF_T(tL0);
	PROF_ADD(A, 2, tM0, tL0);
	fused_publish(LS, seg, lane, totalMeshes, totV, totI);
	uint64_t baseM = 0, baseV = 0, baseI = 0;
	if (!fused_lookback(LS, seg, numSegments, lane, &baseM, &baseV, &baseI)) { fused_fail(A.totals, VGX_E_INTERNAL, VGX_FAIL_LOOKBACK_TIMEOUT, seg, 0); failed = true; }
	PROF_T(tE0);
	PROF_ADD(A, 3, tL0, tE0);
	if (baseV + totV > A.caps.vertices || baseI + totI > A.caps.indices || baseM + totalMeshes > A.caps.meshes) {
		fused_fail(A.totals, VGX_E_NOSPACE, VGX_FAIL_OUT_CAPACITY, seg, (baseV + totV > A.caps.vertices ? 1u : 0u) | (baseI + totI > A.caps.indices ? 2u : 0u) | (baseM + totalMeshes > A.caps.meshes ? 4u : 0u));
		failed = true;
	}
	wt.polyVerts += sc.polyVerts; wt.numSubs += sc.numSubs; wt.numSerial += sc.numSerial; wt.elems += totF + totS; wt.fillElems += totF;
	if (lane == 0) { // batch totals
		if (seg + 1 == numSegments) {
			A.totals->sizes.num_meshes = baseM + totalMeshes;
			A.totals->sizes.num_vertices = baseV + totV;
			A.totals->sizes.num_indices = baseI + totI;
		}
	}
	if (failed || A.totals->status != VGX_OK) { return; }

	// ---- E: mesh table, fills, strokes ------------------------------------------------------------------------------
	if ((uint32_t)lane < totalMeshes) {
		s_mesh[lane].vOff = vOff; s_mesh[lane].iOff = iOff;
		if (A.meshes_out) {
			vgx_mesh r;
			r.first_vertex = baseV + vOff; r.first_index = baseI + iOff;
			r.num_vertices = nv; r.num_indices = ni;
			r.draw = s_mesh[lane].draw;
			r.subpath_kind = (s_mesh[lane].sub & 0x0FFFFFFFu) | ((kind & 0xFu) << 28);
			A.meshes_out[baseM + (uint64_t)lane] = r;
		}
	}
	__syncthreads();

	for (uint32_t chunk = 0; chunk < totF; chunk += VGX_WAVE) { // convex fills
		const uint32_t e = chunk + (uint32_t)lane;
		const bool valid = e < totF;
		const uint32_t wrel = preF <= chunk ? 0u : (preF - chunk > 64u ? 64u : preF - chunk);
		const int k = window_owner_rel(wrel, valid ? (uint32_t)lane : 0u);
		const uint32_t ownerBase = (uint32_t)__shfl((int)preF, k);
		const FMesh r = s_mesh[k];
		FillFetch F;
		const uint32_t j = valid ? e - ownerBase : 0u;
		F.valid = valid;
		F.ibase = 0;
		F.aaElem = valid && VGX_MD_KIND(r.kind) == VGX_MESH_FILL_AA;
		F.prevInWave = lane > 0 && j > 0;
		F.nextInWave = lane < VGX_WAVE - 1 && j + 1 < r.N && e + 1 < totF;
		const VS v = vs_offset(polyBase, r.polyFirst);
		F.p1 = v2(0.0f, 0.0f); F.pNextB = F.p1; F.pPrevB = F.p1;
		if (valid) { F.p1 = v.ld(j); }
		if (F.aaElem && !F.nextInWave) { F.pNextB = v.ld(j + 1 < r.N ? j + 1 : 0); }
		if (F.aaElem && !F.prevInWave) { F.pPrevB = v.ld(j > 0 ? j - 1 : r.N - 1); }
		F.j = j; F.N = r.N; F.color = r.color; F.aa = r.hsw;
		F.firstV = baseV + r.vOff; F.firstI = baseI + r.iOff; F.mi = 0;
		fill_emit_chunk(A.pos, A.color, A.idx, F);
	}

	PROF_T(tS0);
	PROF_ADD(A, 4, tE0, tS0);
	StrokeCarry carry;
	carry.v = 0; carry.i = 0; carry.rails = 0;
	for (uint32_t chunk = 0; chunk < totS; chunk += VGX_WAVE) { // polyline strokes
		const uint32_t e = chunk + (uint32_t)lane;
		const bool valid = e < totS;
		const uint32_t wrel = preS <= chunk ? 0u : (preS - chunk > 64u ? 64u : preS - chunk);
		const int k = window_owner_rel(wrel, valid ? (uint32_t)lane : 0u);
		const uint32_t ownerBase = (uint32_t)__shfl((int)preS, k);
		FMesh r = s_mesh[k];
		if (!valid) { r.N = 2; r.kind = VGX_MESH_STROKE_AA; r.hsw = 0.0f; r.hswAA = 0.0f; r.fringe = 1.0f; r.polyFirst = 0; r.draw = 0; r.vOff = 0; r.iOff = 0; r.color = 0; }
		const MeshCtxT<VS> mc = mesh_ctx_from(r, A.draws, polyBase, valid ? e - ownerBase : 0u);
		const int nvalid = (int)(totS - chunk < (uint32_t)VGX_WAVE ? totS - chunk : (uint32_t)VGX_WAVE);
		const uint64_t firstV = baseV + r.vOff, firstI = baseI + r.iOff;
		stroke_chunk(valid, lane < VGX_WAVE - 1 && e + 1 < totS, nvalid, lane, mc, r.color, A.pos + 2 * firstV, A.color + firstV, A.idx + firstI, 0u, carry);
	}
	PROF_T(tS1);
	PROF_ADD(A, 5, tS0, tS1);
	PROF_INC(A, 6, 1);
}

__global__ __launch_bounds__(VGX_WAVE) void k_tess_fused(VgxFusedArgs A)
{
	__shared__ vgx_f2 s_poly[FUSED_P];
	__shared__ __attribute__((aligned(16))) unsigned char s_union[FUSED_UNION_BYTES];
	__shared__ VgxSegSub s_sub[FUSED_T];
	__shared__ FDraw s_draw[FUSED_T];
	__shared__ uint32_t s_res[SR_COUNT + 1];
	float2* s_stack = (float2*)s_union;
	float2* s_leaf = s_stack + FUSED_STACK_F2;
	FMesh* s_mesh = (FMesh*)s_union;
	const int lane = threadIdx.x;
	if (A.totals->status != VGX_OK) { return; } // validation / capacity errors of the command-prefix scan
	const uint64_t totalCmds = A.cmd_prefix[A.ndraws];
	const uint64_t segItems = A.seg_items;
	const uint64_t numSegments = (totalCmds + segItems - 1) / segItems;
	if (numSegments > A.seg_cap) { return; } // k_seg_starts reported VGX_E_NOSPACE
	WaveTotals wt;
	wt.polyVerts = 0; wt.numSubs = 0; wt.numSerial = 0; wt.elems = 0; wt.fillElems = 0;

	for (;;) {
		PROF_T(tT0);
		uint32_t t = 0;
		if (lane == 0) { t = atomicAdd(A.ticket, (uint32_t)VGX_FUSED_TICKET); }
		t = wave_bcast_u32(t, 0);
		PROF_T(tT1);
		PROF_ADD(A, 0, tT0, tT1);
		if ((uint64_t)t >= numSegments) { break; }
		const uint64_t segEnd = ((uint64_t)t + VGX_FUSED_TICKET < numSegments) ? (uint64_t)t + VGX_FUSED_TICKET : numSegments;
		for (uint64_t seg = t; seg < segEnd; ++seg) {
			const uint64_t d0 = A.seg_start[seg], d1 = A.seg_start[seg + 1];
			const uint64_t C0 = A.cmd_prefix[d0], C1 = A.cmd_prefix[d1];
			uint32_t nd = (uint32_t)(d1 - d0);
			bool failed = false;
			if (d1 - d0 > FUSED_T) { failed = true; nd = 0; fused_fail(A.totals, VGX_E_NOSPACE, VGX_FAIL_SEG_DRAWS, seg, (uint32_t)(d1 - d0)); }
			SegCounts sc;
			sc.polyVerts = 0; sc.numSubs = 0; sc.subRecs = 0; sc.numSerial = 0;
			bool toHeap = false;
			float2* heap = nullptr;
			PROF_T(tF0);
			if (!failed) {
				// the segment's draws, one per lane
				FusedWindow W;
				{
					const uint64_t idx = d0 + (uint64_t)lane;
					W.prefix = (idx <= A.ndraws) ? A.cmd_prefix[idx] : ~0ull;
					W.pc0 = 0; W.serial = 0;
					if (idx < d1) {
						const uint32_t path = A.draws[idx].path;
						W.pc0 = A.ps.path_cmd_begin[path];
						W.serial = A.ps.path_flags[path] & VGX_PF_SERIAL;
					}
				}
				s_draw[lane] = FDraw{0, 0, 0, 0};
				const bool staticSerial = wave_ballot(W.serial != 0) != 0;
				FlatResult R;
				R.anySerial = staticSerial; R.overflowP = false; R.overflowT = false; R.cursor = 0; R.polyTrue = 0; R.subRecs = 0; R.numSubs = 0;
				if (!staticSerial) {
					R = fused_flatten(A, d0, C0, C1, W, lane, s_poly, s_stack, s_leaf, s_sub, s_draw, false, nullptr, 0);
				} else if ((uint32_t)lane < nd) {
					s_draw[lane].serial = W.serial;
				}
				if (R.anySerial) {
					__syncthreads(); // the serial flags in s_draw
					fused_serial_segment(A.draws, A.ps.path_cmd_begin, A.ps.cmd_type, A.ps.cmd_arg_off, A.ps.args, A.heap, A.heap_cap, A.totals, d0, nd, (float*)s_poly, s_sub, s_draw, s_res);
					__syncthreads();
					sc.polyVerts = s_res[SR_POLY]; sc.numSubs = s_res[SR_NUMSUBS]; sc.subRecs = s_res[SR_SUBRECS]; sc.numSerial = s_res[SR_NUMSERIAL];
					const uint32_t fl = s_res[SR_FLAGS];
					toHeap = (fl & SRF_TOHEAP) != 0;
					heap = (float2*)A.heap + (((uint64_t)s_res[SR_HEAP_HI] << 32) | s_res[SR_HEAP_LO]);
					if (fl & (SRF_FAILED | SRF_OVERFLOW_T)) { failed = true; fused_fail(A.totals, VGX_E_NOSPACE, (fl & SRF_FAILED) ? VGX_FAIL_SERIAL_HEAP : VGX_FAIL_SERIAL_SUBRECS, seg, sc.subRecs); }
				} else {
					if (R.overflowT) { failed = true; fused_fail(A.totals, VGX_E_NOSPACE, VGX_FAIL_SEG_SUBRECS, seg, R.subRecs); }
					else if (R.overflowP) { // does not fit the LDS window: once more, into a block of the polyline heap
						// + VGX_WAVE: inside a chunk the running position overshoots the chunk's net total by the vertices its
						// pathClose pops remove again (at most one per two lanes)
						const uint32_t want = (uint32_t)R.cursor + VGX_WAVE;
						unsigned long long base = 0;
						if (lane == 0) { base = atomicAdd(&A.totals->poly_heap_cursor, (unsigned long long)want); }
						base = wave_bcast_u64(base, 0);
						if (base + (uint64_t)want > A.heap_cap) { failed = true; fused_fail(A.totals, VGX_E_NOSPACE, VGX_FAIL_HEAP, seg, want); }
						else {
							toHeap = true;
							heap = (float2*)A.heap + base;
							s_draw[lane] = FDraw{0, 0, 0, 0};
							R = fused_flatten(A, d0, C0, C1, W, lane, s_poly, s_stack, s_leaf, s_sub, s_draw, true, heap, want);
						}
					}
					sc.polyVerts = (uint32_t)(R.polyTrue > 0 ? R.polyTrue : 0); sc.numSubs = R.numSubs; sc.subRecs = R.subRecs;
				}
				__syncthreads(); // window / tables written by all lanes, read by all lanes
			}
			PROF_T(tF1);
			PROF_ADD(A, 1, tF0, tF1);
			if (toHeap) {
				__threadfence_block();
				VtxGlobal vs; vs.p = heap;
				fused_finish<VtxGlobal>(A, seg, numSegments, d0, nd, sc, failed, lane, vs, s_sub, s_draw, s_mesh, wt);
			} else {
				VtxLds vs; vs.p = (vgx_lds_cf2p)s_poly;
				fused_finish<VtxLds>(A, seg, numSegments, d0, nd, sc, failed, lane, vs, s_sub, s_draw, s_mesh, wt);
			}
			__syncthreads(); // the next segment overwrites window and tables
		}
	}
	if (lane == 0) {
		if (wt.polyVerts) { atomicAdd((unsigned long long*)&A.totals->sizes.num_poly_vertices, (unsigned long long)wt.polyVerts); }
		if (wt.numSubs) { atomicAdd((unsigned long long*)&A.totals->sizes.num_subpaths, (unsigned long long)wt.numSubs); }
		if (wt.numSerial) { atomicAdd((unsigned long long*)&A.totals->sizes.num_serial_draws, (unsigned long long)wt.numSerial); }
		if (wt.elems) { atomicAdd((unsigned long long*)&A.totals->sizes.num_elements, (unsigned long long)wt.elems); }
		if (wt.fillElems) { atomicAdd((unsigned long long*)&A.totals->sizes.num_fill_elements, (unsigned long long)wt.fillElems); }
	}
}

// First draw of every segment: seg_start[k] = first draw d with cmd_prefix[d] >= k * seg_items (or ndraws), for
// k = 0 .. numSegments. One thread per draw writes the (usually zero or one) segment boundaries that fall on it.
__global__ __launch_bounds__(256) void k_seg_starts(VgxFusedArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	const uint64_t nd = A.ndraws;
	const uint64_t total = A.cmd_prefix[nd];
	const uint64_t s = A.seg_items;
	const uint64_t numSegments = (total + s - 1) / s;
	if (numSegments > A.seg_cap) { // more segments than the tables vgx_tessellate_count sized: a batch unlike the counted one
		if (blockIdx.x == 0 && threadIdx.x == 0) { fused_fail(A.totals, VGX_E_NOSPACE, VGX_FAIL_SEG_TABLE, numSegments, 0); }
		return;
	}
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nd; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t kLo = (i == 0) ? 0 : A.cmd_prefix[i - 1] / s + 1;
		uint64_t kHi = (i < nd) ? A.cmd_prefix[i] / s : numSegments;
		if (kHi > numSegments) { kHi = numSegments; }
		for (uint64_t k = kLo; k <= kHi; ++k) { A.seg_start[k] = i; }
	}
}

// vgx_tessellate_count: would the fused kernel's tables hold every segment at bucket size seg_items[c]? Draw i belongs to
// segment floor(cmd_prefix[i] / s); the thread of a segment's first draw walks the segment. out[4 * c + 0] non-empty
// segments, + 1 table violations (more than VGX_FUSED_T draws or meshes), + 2 segments whose polyline exceeds the LDS
// window (they take the heap path: slow, legal).
__global__ __launch_bounds__(256) void k_fused_probe(const uint64_t* cmd_prefix, const vgx_draw_info* dinfo, uint64_t ndraws, VgxFusedProbe P)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndraws; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t p = cmd_prefix[i];
		const uint64_t pPrev = (i == 0) ? 0 : cmd_prefix[i - 1];
		for (int c = 0; c < VGX_FUSED_CANDIDATES; ++c) {
			const uint64_t s = P.seg_items[c];
			const uint64_t k = p / s;
			if (i != 0 && pPrev / s == k) { continue; } // not the first draw of its segment
			uint64_t nd = 0, meshes = 0, poly = 0;
			for (uint64_t j = i; j < ndraws && cmd_prefix[j] / s == k; ++j) {
				++nd; meshes += dinfo[j].num_meshes; poly += dinfo[j].num_poly_vertices;
				if (nd > VGX_FUSED_T) { break; }
			}
			atomicAdd((unsigned long long*)&P.out[4 * c + 0], 1ull);
			if (nd > VGX_FUSED_T || meshes > VGX_FUSED_T) { atomicAdd((unsigned long long*)&P.out[4 * c + 1], 1ull); }
			if (poly > VGX_FUSED_P) { atomicAdd((unsigned long long*)&P.out[4 * c + 2], 1ull); }
		}
	}
}

} // namespace

void vgx_launch_fused(const VgxFusedArgs& a, int waves, hipStream_t s)
{
	hipLaunchKernelGGL(k_seg_starts, dim3(1024), dim3(256), 0, s, a);
	hipLaunchKernelGGL(k_tess_fused, dim3(waves), dim3(VGX_WAVE), 0, s, a);
}

void vgx_launch_fused_probe(const uint64_t* cmd_prefix, const vgx_draw_info* dinfo, uint64_t ndraws, const VgxFusedProbe& p, hipStream_t s)
{
	hipLaunchKernelGGL(k_fused_probe, dim3(2048), dim3(256), 0, s, cmd_prefix, dinfo, ndraws, p);
}
