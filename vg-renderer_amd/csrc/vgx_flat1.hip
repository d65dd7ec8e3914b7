// vgx_flat1.hip -- ordered ONE-WALK flatten on gfx950 (vgx_flatten; replaces vg::pathXXX, reference src/path.cpp).
//
// The two-pass kernels of vgx_flatten.hip (k_flatten<count> -> scan over draws -> k_flatten<emit>) walk every cubic twice, keep
// one word per command instance in HBM between the passes and store a cubic's leaves four at a time from a lane that is
// 360 bytes away from its neighbours' (round 4: 2.1x write amplification, 0.08 of the HBM peak). This kernel produces the same
// ORDERED output -- draws concatenated in order, pathGetVertices / pathGetSubPaths per draw -- from one walk:
//
//   segment   = the draws whose first command instance falls into one bucket of `segItems` command instances (whole draws, as
//               in vgx_flatten.hip; segItems = 64 minus the batch's mean draw length, so that most segments are ONE 64-command
//               chunk). Segments are handed out by a ticket counter: ticket order = output order = execution order.
//   tasks     = the chunk's cubics, compacted: a cubic whose root is not flat is cut at its root into two TASKS (left and
//               right half) while the wave has free lanes, so the MOVE_TO / LINE_TO / CLOSE lanes of a chunk walk too (1 M
//               moveTo + cubicTo paths: 32 cubics -> 64 tasks per wave). One lane walks one task depth first
//               (path.cpp:104-181: same expressions, same order), pending right halves in LDS.
//   leaves    = appended to ONE list in LDS in the order they are found (ballot + popcount per step), tagged (task, k).
//               "The growing polyline staged in LDS": a chunk's ~1500 vertices never exist in HBM in any other order.
//   offsets   = the chunk's prefix scans (vgx_wave.h, segmented by draw / sub-path) give every command its place inside the
//               chunk; the segment's place in the output comes from a decoupled look-back over per-segment totals
//               (vertices, sub-paths, meshes): no count pass, no scan kernel, no per-command words.
//   stores    = lane i of the wave moves list entry i to out[base + place(task) + k]: a chunk's output is one contiguous
//               ~12 KB range that the wave writes in one burst (every 64-byte line is completed within a few instructions).
//
// Segments with more than one chunk (a draw longer than 64 commands, or several medium-sized draws that begin in one bucket)
// cannot publish their total after one chunk: they count all their chunks first (same walk, nothing staged), publish, and walk
// again chunk by chunk -- the old cost, for those segments only. A chunk whose leaves do not fit the list (CAP) is placed by
// walking its cubics again, straight to memory.
//
// Exactness: as in vgx_flatten.hip. A draw that hits pathAddVertex's epsilon de-duplication (path.cpp:767-777) or the silent
// drop at stack depth 10 (path.cpp:168-179), and any path with ARC / ARC_TO / closed shapes, is done by ONE lane running the exact
// sequential builder (k_flatten_serial): statically serial paths are counted BEFORE this kernel, a draw found degenerate DURING
// it is listed, counted, and the kernel runs a second time (pass 1) knowing it. Degenerate input is slow, never wrong.
#include <stdlib.h>
#include "vgx_internal.h"
#include "vgx_wave.h"
#include "vgx_pathsim.h"
#include "vgx_walk.h"
#include "vgx_flat1.h"

namespace {

// -DVGX_F1_PROFILE: shader-clock ticks per phase of k_flat1 summed over all waves into VgxTotals::prof (read through vgx_get_failure_info)
#ifdef VGX_F1_PROFILE
__device__ unsigned long long* g_f1_dbg = nullptr; // [tickets][4]: wall clock (100 MHz) at the ticket / at A(t) / after the look-back, hardware id
__device__ unsigned long long g_f1_dbg_n = 0;
#define F1_DBG(t, i, v) do { if (g_f1_dbg && lane == 0 && (t) < g_f1_dbg_n) { g_f1_dbg[(t) * 4 + (i)] = (v); } } while (0)
#define F1_WALL() ((unsigned long long)wall_clock64())
#define F1_CLK() ((unsigned long long)clock64())
#define F1_ACC(i, v) (prof##i += (v))
#else
#define F1_DBG(t, i, v) ((void)(v))
#define F1_WALL() 0ull
#define F1_CLK() 0ull
#define F1_ACC(i, v) ((void)(v))
#endif

// Registers: the leaf list leaves room for 1.5 waves per SIMD; the kernel must not need more registers than two waves per SIMD get.
#ifndef F1_MIN_WAVES_PER_EU
#define F1_MIN_WAVES_PER_EU 2
#endif
#ifndef VGX_F1_LV
#define VGX_F1_LV 6 /* pending right halves per lane kept in LDS by the hot loop; a task that nests deeper is redone with all 10 */
#endif
static_assert(2 * VGX_F1_LV >= VGX_CUBIC_MAX_PENDING, "the full-depth redo spreads one lane's stack over two columns");

// Full-depth pending stack without private memory: levels [0, LV) in column A, [LV, 10) in column B of the hot loop's LDS
// stack. Only the rare redo of a deeply nested / degenerate cubic uses it, 32 lanes at a time.
template<int LV>
struct LdsStack2
{
	float2* a; float2* b;
	__device__ __forceinline__ float2* at(int level) const { return level < LV ? a + level * 3 * VGX_WAVE : b + (level - LV) * 3 * VGX_WAVE; }
	__device__ __forceinline__ void push(int level, float ax, float ay, float bx, float by, float cx, float cy)
	{
		float2* p = at(level);
		p[0] = make_float2(ax, ay); p[VGX_WAVE] = make_float2(bx, by); p[2 * VGX_WAVE] = make_float2(cx, cy);
	}
	__device__ __forceinline__ void pop(int level, float& ax, float& ay, float& bx, float& by, float& cx, float& cy)
	{
		const float2* p = at(level);
		const float2 q0 = p[0], q1 = p[VGX_WAVE], q2 = p[2 * VGX_WAVE];
		ax = q0.x; ay = q0.y; bx = q1.x; by = q1.y; cx = q2.x; cy = q2.y;
	}
};

// ---- look-back records: two words per segment, each carries its own state in bits 62-63 -----------------------------
//   w[0] = state << 62 | polyline vertices        w[1] = state << 62 | sub-paths << 31 | meshes   (each < 2^31, checked)
#define F1_A 1ull /* aggregate: the segment's own totals */
#define F1_P 2ull /* inclusive prefix: everything up to and including the segment */
#define F1_VAL ((1ull << 62) - 1ull)
#define F1_SM_LIMIT (1ull << 31)

// A capacity verdict does not end the call: the segments are still counted (nothing is written), so that the totals tell the caller
// what the batch needs. Anything else does.
__device__ __forceinline__ bool f1_fatal(uint32_t st) { return st != VGX_OK && st != VGX_E_NOSPACE; }

__device__ __forceinline__ void f1_publish(VgxF1Seg* segs, uint64_t t, uint64_t state, uint64_t v, uint64_t s, uint64_t m)
{
	__hip_atomic_store(&segs[t].w[0], (state << 62) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(&segs[t].w[1], (state << 62) | (s << 31) | m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) { v += (uint64_t)__shfl_xor((unsigned long long)v, o); }
	return v;
}

// Sums of all segments in front of `t`: TWO levels. Waves run in step (same work per segment, ~1300 of them in flight), so when a
// segment is counted none of the ~1300 in front of it knows its prefix yet: a flat look-back walked back through all of them,
// 64 (then 256) records per memory round trip -- 80 000 of a segment's 190 000 cycles. Here 64 consecutive tickets form a GROUP:
//   ticket record  A(t)   the segment's own totals
//   group record   GA(g)  the group's totals, published by the wave that holds the group's LAST ticket as soon as it has seen
//                         the other 63 aggregates (they are its own look-back window);  GP(g) = inclusive prefix, same wave
// and the sums in front of ticket t = (g, r) are  A(t - r .. t - 1)  +  GA(g - 1), GA(g - 2), ... back to the nearest GP:
// one request of <= 63 ticket records and <= 64 group records (4096 tickets), both in flight together.
// Wave-uniform result; false = gave up (another wave reported an error, or the bounded wait ran out: status set).
__device__ __forceinline__ bool f1_wait_more(uint32_t* spins, VgxTotals* totals, uint64_t t, int lane)
{
	if ((++*spins & 63u) == 0u) {
		const uint32_t st = __hip_atomic_load(&totals->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (f1_fatal(st)) { return false; }
		if (*spins > (1u << 22)) { // ~ seconds: cannot happen while tickets are handed out in order; never hang the device
			if (lane == 0) {
				if (atomicCAS(&totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_INTERNAL) == VGX_OK) { totals->fail_reason = VGX_FAIL_LOOKBACK_TIMEOUT; totals->fail_segment = t; }
			}
			return false;
		}
	}
	__builtin_amdgcn_s_sleep(1);
	return true;
}

__device__ __forceinline__ void f1_sum2(uint64_t v, uint64_t sm, uint64_t* av, uint64_t* as, uint64_t* am)
{
	*av += wave_sum_u64(v);
	// sub-paths and meshes travel as two 31-bit fields: summed apart (64 records could carry into the neighbour)
	*as += wave_sum_u64(sm >> 31);
	*am += wave_sum_u64(sm & (F1_SM_LIMIT - 1));
}

// ---- the NEXT segment's front, requested while this segment waits for its place ---------------------------------------------
// A segment's first four memory round trips (ticket -> segment table -> command prefix / draw window -> the window's paths) as a
// little state machine, one dependent stage per call. Two ways of running it in the shadow of the previous segment were measured
// and NOT kept: (a) one stage per poll of the look-back -- a ticket taken before the look-back's wait is worked on only after that
// wait, whose length varies from wave to wave, so the segment arrives late for its place in the order and the waves behind it
// wait even longer: 0.90 -> 1.66 ms; (b) one stage every other step of the placement loop -- every commit waits for the
// scattered stores issued before it (vmcnt counts loads and stores in order): placement 13 600 -> 31 000 cycles per segment,
// 0.90 -> 1.07 ms. The stages run back to back at the top of a segment.
struct F1Next
{
	uint32_t stage;   // 0 nothing; 1 ticket taken; 2 + segment table; 3 + command prefix and the window's prefix / path; 4 complete
	unsigned long long t;
	uint64_t d0, d1, C0, C1;
	DrawWindow W;     // window over draws [d0, d0 + 64)
	uint32_t wpath;
};

struct F1NextRaw { unsigned long long a, b, c; uint32_t u, v; };

// issue the loads of the next stage (nothing of their results is used here)
__device__ __forceinline__ F1NextRaw f1_next_issue(const F1Next& N, const VgxFlattenArgs& A, const VgxF1Args& X, VgxTotals* T, int lane, uint64_t numSegments)
{
	F1NextRaw R; R.a = 0; R.b = 0; R.c = 0; R.u = 0; R.v = 0;
	if (N.stage == 0) {
		if (lane == 0) { R.a = atomicAdd(&T->flat_ticket, 1ull); }
	} else if (N.stage == 1) {
		if (N.t < numSegments) { R.a = X.seg_draw[N.t]; R.b = X.seg_draw[N.t + 1]; }
	} else if (N.stage == 2) {
		if (N.d0 < N.d1) { R.a = A.cmd_prefix[N.d0]; R.b = A.cmd_prefix[N.d1]; }
		const uint64_t idx = N.d0 + (uint64_t)lane;
		R.c = (idx <= A.ndraws) ? A.cmd_prefix[idx] : ~0ull;
		if (idx < A.ndraws) { R.u = A.draws[idx].path; }
	} else if (N.stage == 3) {
		const uint64_t idx = N.d0 + (uint64_t)lane;
		if (idx < A.ndraws) { R.u = A.ps.path_cmd_begin[N.wpath]; R.v = A.ps.path_flags[N.wpath] & (VGX_PF_SERIAL | VGX_PF_THIN); }
	}
	return R;
}

__device__ __forceinline__ void f1_next_commit(F1Next& N, const F1NextRaw& R, uint64_t numSegments)
{
	if (N.stage == 0) { N.t = wave_bcast_u64(R.a, 0); N.stage = N.t < numSegments ? 1u : 4u; }
	else if (N.stage == 1) { N.d0 = R.a; N.d1 = R.b; N.C0 = 0; N.C1 = 0; N.stage = 2; }
	else if (N.stage == 2) { if (N.d0 < N.d1) { N.C0 = R.a; N.C1 = R.b; } N.W.prefix = R.c; N.wpath = R.u; N.W.pc0 = 0; N.W.serial = 0; N.stage = 3; }
	else if (N.stage == 3) { N.W.pc0 = R.u; N.W.serial = R.v; N.stage = 4; }
}

// myV / myS / myM: the segment's own totals (already published as A(t)); lastOfGroup: this wave also publishes the group records.
// A record that was seen published is not read again: a waiting wave re-reads only what it still waits for (1300 waves re-reading
// their whole windows every microsecond made the few memory channels that hold the live records the bottleneck of the kernel).
__device__ __forceinline__ bool f1_lookback(const VgxF1Seg* segs, VgxF1Seg* grps, uint64_t t, bool lastOfGroup, uint64_t myV, uint64_t myS, uint64_t myM,
	int lane, VgxTotals* totals, uint64_t* bv, uint64_t* bs, uint64_t* bm, unsigned long long* profSpins = nullptr, unsigned long long* profFirst = nullptr)
{
	const uint64_t g = t >> 6;
	const int r = (int)(t & 63ull);
	uint64_t inV = 0, inS = 0, inM = 0;   // tickets of my group in front of me
	uint64_t gV = 0, gS = 0, gM = 0;      // groups in front of mine
	uint32_t spins = 0;
	bool inDone = r == 0, grpDone = g == 0;
	uint64_t ghi = g; // groups [0, ghi) are still to be summed
	uint64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
	bool gotA = false, gotB = false; // my lane's record has been seen published
	const unsigned long long c0 = F1_CLK();
	bool first = true;
	while (!inDone || !grpDone) {
		const bool haveA = !inDone && lane < r;
		const bool haveB = !grpDone && (uint64_t)lane < ghi;
		if (haveA && !gotA) {
			a0 = __hip_atomic_load(&segs[t - 1 - (uint64_t)lane].w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			a1 = __hip_atomic_load(&segs[t - 1 - (uint64_t)lane].w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			gotA = (a0 >> 62) == (a1 >> 62) && (a0 >> 62) != 0; // the two words of one publication
		}
		if (haveB && !gotB) {
			b0 = __hip_atomic_load(&grps[ghi - 1 - (uint64_t)lane].w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			b1 = __hip_atomic_load(&grps[ghi - 1 - (uint64_t)lane].w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			gotB = (b0 >> 62) == (b1 >> 62) && (b0 >> 62) != 0;
		}
		bool progress = false;
		if (!inDone) {
			const uint64_t rm = wave_ballot(!haveA || gotA);
			if (first && profFirst) { *profFirst += F1_CLK() - c0; }
			if (rm == ~0ull) {
				f1_sum2(haveA ? (a0 & F1_VAL) : 0ull, haveA ? (a1 & F1_VAL) : 0ull, &inV, &inS, &inM);
				inDone = true; progress = true;
				if (lastOfGroup && lane == 0) { f1_publish(grps, g, F1_A, inV + myV, inS + myS, inM + myM); } // GA(g): the groups behind need not wait for my prefix
			}
		}
		first = false;
		if (!grpDone) {
			const uint64_t pm = wave_ballot(haveB && gotB && (b0 >> 62) == F1_P);
			const uint64_t rm = wave_ballot(!haveB || gotB);
			const int fp = pm ? __builtin_ctzll(pm) : 64;
			const uint64_t need = fp >= 63 ? ~0ull : lanemask_le(fp);
			if ((rm & need) == need) {
				const bool take = haveB && lane <= fp;
				f1_sum2(take ? (b0 & F1_VAL) : 0ull, take ? (b1 & F1_VAL) : 0ull, &gV, &gS, &gM);
				progress = true;
				if (fp < 64 || ghi <= 64) { grpDone = true; } else { ghi -= 64; gotB = false; } // (no known prefix within 4096 tickets: further back)
			}
		}
		if (!progress && !f1_wait_more(&spins, totals, t, lane)) { return false; }
	}
	if (profSpins) { *profSpins += spins; }
	if (lastOfGroup && r == 0 && lane == 0) { f1_publish(grps, g, F1_A, myV, myS, myM); } // a group of one ticket (the batch's last)
	if (lastOfGroup && lane == 0) { f1_publish(grps, g, F1_P, gV + inV + myV, gS + inS + myS, gM + inM + myM); }
	*bv = gV + inV; *bs = gS + inS; *bm = gM + inM;
	return true;
}

// ---- LDS of one wave ---------------------------------------------------------------------------------------------------
// stack  [LV * 3][64] float2   pending right halves of the task walks, lane-interleaved
// list   [CAP] float2          leaves in the order they were found (untransformed)
// tag    [CAP] uint16          task | k << 6 (k = rank of the leaf inside its task, < 1024: a task has at most 2^10 leaves)
// tinfo  [64] uint2            per task: leaf count | flags (after the walk), then place | limit (for the placement)
// The task records (control points, tolerance) of the walk's start alias the list: they are read before the first leaf lands.
#define F1_TF_SLOW 0x40000000u
#define F1_TF_ABORT 0x80000000u
#define F1_PARAM_WORDS 10

template<int CAP>
struct F1Lds
{
	float2 stack[VGX_F1_LV * 3 * VGX_WAVE];
	float2 list[CAP + 2];          // + the dummy entry lanes without a leaf write to (and padding to 16 bytes)
	unsigned short tag[CAP + 8];
	uint2 tinfo[VGX_WAVE];
};
static_assert(2 * F1_PARAM_WORDS * VGX_WAVE * 4 <= 1024 * 8, "the two task-record buffers alias the leaf list");

// One task walk per lane. Same arithmetic as build_flatten_hot (vgx_walk.h) = path.cpp:105-129; leaves go to the shared list
// when `stage`. Returns the number of list entries the wave appended in total (it may exceed CAP: nothing is stored past it).
//
// The loop body is straight-line code -- no exec-mask branch, no LDS round trip on its critical path (the first version, with
// the compiler's branches around push / leaf / pop and the pop's three ds_reads waited for on the spot, took 880 cycles per step
// at the 1.5 waves per SIMD the leaf list leaves room for):
//   pop   the top entry of the pending stack is REQUESTED at the top of every step and consumed at its end if the node turns out
//         flat -- the LDS latency runs beside the step's arithmetic;
//   push  the right half is WRITTEN every step to the free row above the top: harmless when the node is not split (nothing reads
//         that row before the next push overwrites it; with all LV rows in use the row being popped is overwritten instead);
//   leaf  lanes without a leaf write theirs to a dummy entry behind the list.
template<int CAP>
__device__ __forceinline__ uint32_t f1_task_walk(F1Lds<CAP>& L, int lane, bool active, v2f P1, v2f P2, v2f P3, v2f P4, float tessTol, bool stage,
	uint32_t* kOut, bool* slowOut, bool* abortOut, unsigned long long* stepsOut = nullptr)
{
	unsigned long long steps = 0;
	float2* stackLane = &L.stack[lane];
	const uint32_t ROW = 3 * VGX_WAVE;
	uint32_t sp = 0, k = 0, listN = 0; // sp = pending * ROW
	bool slow = false;
	bool more = active, aborted = false;
	while (wave_ballot(more)) {
		// request the top pending entry (row pending - 1; an empty stack reads row 0: not used)
		const uint32_t rp = sp >= ROW ? sp - ROW : 0u;
		const float2 q2 = stackLane[rp], q3 = stackLane[rp + VGX_WAVE], q4 = stackLane[rp + 2 * VGX_WAVE];
		const v2f d = P4 - P1;
		const v2f a2 = P2 - P4, a3 = P3 - P4;
		const v2f dsw = d.yx;
		const v2f m2 = a2 * dsw, m3 = a3 * dsw;
		const float d2 = __builtin_fabsf(m2.x - m2.y), d3 = __builtin_fabsf(m3.x - m3.y);
		const float d23 = d2 + d3;
		const v2f dd = d * d;
		const float len2 = dd.x + dd.y;
		const bool flat = d23 * d23 <= tessTol * len2;
		const v2f P12 = (P1 + P2) * 0.5f, P23 = (P2 + P3) * 0.5f, P34 = (P3 + P4) * 0.5f;
		const v2f P123 = (P12 + P23) * 0.5f, P234 = (P23 + P34) * 0.5f;
		const v2f P1234 = (P123 + P234) * 0.5f;
		const bool full = sp >= (uint32_t)VGX_F1_LV * ROW;
		const bool push = more && !flat && !full;
		const bool isLeaf = more && flat;
		const bool pop = isLeaf && sp != 0u;
		aborted = aborted || (more && !flat && full); // nests deeper than the LDS levels (or past the reference's depth limit): the owner redoes the cubic
		// the right half, to the free row (the row being popped / abandoned when all rows are in use)
		const uint32_t wp = full ? sp - ROW : sp;
		stackLane[wp] = make_float2(P234.x, P234.y);
		stackLane[wp + VGX_WAVE] = make_float2(P34.x, P34.y);
		stackLane[wp + 2 * VGX_WAVE] = make_float2(P4.x, P4.y);
		// the leaf, to the list
		const uint64_t leafMask = wave_ballot(isLeaf);
		uint32_t li = listN + (uint32_t)__popcll(leafMask & lanemask_lt(lane));
		li = (isLeaf && li < (uint32_t)CAP) ? li : (uint32_t)CAP;
		if (stage) { // wave-uniform
			L.list[li] = make_float2(P4.x, P4.y);
			L.tag[li] = (unsigned short)((uint32_t)lane | (k << 6));
		}
		// pathAddVertex's epsilon test (path.cpp:769-775) compares a leaf with the previous vertex = the flat piece's own first
		// point: the squared length the flatness test just computed ((P1 - P4)^2 and (P4 - P1)^2 are the same floats)
		slow = slow || (isLeaf && len2 < VGM_EPSILON);
		k += isLeaf ? 1u : 0u;
		listN += (uint32_t)__popcll(leafMask);
		// next node: the left half, or the popped right half (which starts where this subtree ended)
		P1.x = push ? P1.x : P4.x; P1.y = push ? P1.y : P4.y;
		P2.x = push ? P12.x : q2.x; P2.y = push ? P12.y : q2.y;
		P3.x = push ? P123.x : q3.x; P3.y = push ? P123.y : q3.y;
		P4.x = push ? P1234.x : q4.x; P4.y = push ? P1234.y : q4.y;
		sp = push ? sp + ROW : (pop ? sp - ROW : sp);
		more = push || pop;
		++steps;
	}
	if (stepsOut) { *stepsOut += steps; }
	*kOut = k;
	*slowOut = slow;
	*abortOut = aborted;
	return listN;
}

// What the placement of a chunk needs of one lane's command instance (k_flat1: placeChunk). 31 dwords.
struct F1Pend
{
	uint32_t flags; // bit 0 valid, 1 serialDraw, 2 endNearFirst, 3 isCubic, 4 deep, 5 lastInSub, 6 closedHere, 7 drawHead, 8 drawLast, 9 slowDraw, 10 serialTail
	int excl, rawCnt, spTotal;
	uint32_t myTasks, myTask0, type;
	float ptx, pty;
	uint32_t na, argOff, recIndex;
	float tessTol;
	int spBefore, subInclC, meshInclC, subAll, meshAll;
	uint32_t serV, serS, serM, serF;
	int inDrawBefore, cnt, subsIncl, fillIncl, strokeIncl;
	uint64_t d, ownerBase;
};

// REDO: the second run of a batch in which the first found degenerate draws -- the same code as its own kernel, so that a kernel
// trace shows the run that does the work apart from the one that (normally) exits at once.
template<int CAP, bool XFORM, bool REDO>
__global__ __launch_bounds__(VGX_WAVE, (CAP > 2048 ? 1 : F1_MIN_WAVES_PER_EU)) void k_flat1(VgxFlattenArgs A, VgxF1Args X)
{
	__shared__ __attribute__((aligned(16))) F1Lds<CAP> L;
	const int lane = threadIdx.x;
	const VgxPathSetDev& ps = A.ps;
	VgxTotals* T = A.totals;
	if (REDO && T->flat_redo == 0u) { return; } // the second run is only for batches in which the first found degenerate draws
	const uint64_t totalCmds = A.cmd_prefix[A.ndraws];
	const uint64_t segItems = vgx_f1_segment_items(totalCmds, A.ndraws, X.seg_max);
	const uint64_t numSegments = (totalCmds + segItems - 1) / segItems;
	const bool readFlags = X.read_flags != 0; // some draws may already be marked serial in dinfo (static serial paths counted up front / pass 1)

	uint64_t wbase = 0;
	DrawWindow W;
	W.prefix = ~0ull; W.pc0 = 0; W.serial = 0;
	F1Next N;
	N.stage = 0; N.t = 0; N.d0 = 0; N.d1 = 0; N.C0 = 0; N.C1 = 0; N.W = W; N.wpath = 0;
#ifdef VGX_F1_PROFILE
	unsigned long long prof0 = 0, prof1 = 0, prof2 = 0, prof3 = 0, prof4 = 0, prof5 = 0, prof6 = 0, prof7 = 0, prof8 = 0;
#define F1_FLUSH() do { if (lane == 0) { atomicAdd(&T->prof[0], prof0); atomicAdd(&T->prof[1], prof1); atomicAdd(&T->prof[2], prof2); atomicAdd(&T->prof[3], prof3); \
	atomicAdd(&T->prof[4], prof4); atomicAdd(&T->prof[5], prof5); atomicAdd(&T->prof[6], prof6); atomicAdd(&T->prof[7], prof7); atomicAdd(&T->prof[8], prof8); } } while (0)
#else
#define F1_FLUSH() ((void)0)
#endif

	// ---- placement of ONE chunk (vertices from the leaf list / the commands, sub-path and per-draw records) from what its lanes knew after the
	// bookkeeping (F1Pend) and the places of the segment (bases: the sums in front of it; run*: placed in front of the chunk inside it). A
	// function of its arguments and the LDS list only, so that it can run LATER than the walk that filled the list (round 6: see `defer`).
	auto placeChunk = [&](const F1Pend& q, uint64_t cubicMask, bool staged, uint32_t listN, uint64_t baseV, uint64_t baseS, uint64_t baseM, long long runV, uint64_t runS, uint64_t runM) {
		const bool valid = (q.flags & 1u) != 0, serialDraw = (q.flags & 2u) != 0, endNearFirst = (q.flags & 4u) != 0, isCubic = (q.flags & 8u) != 0, deep = (q.flags & 16u) != 0;
		const bool lastInSub = (q.flags & 32u) != 0, closedHere = (q.flags & 64u) != 0, drawHead = (q.flags & 128u) != 0, drawLast = (q.flags & 256u) != 0, slowDraw = (q.flags & 512u) != 0, serialTail = (q.flags & 1024u) != 0;
		const int excl = q.excl, rawCnt = q.rawCnt, spTotal = q.spTotal, spBefore = q.spBefore, subInclC = q.subInclC, meshInclC = q.meshInclC, subAll = q.subAll, meshAll = q.meshAll;
		const int inDrawBefore = q.inDrawBefore, cnt = q.cnt, subsIncl = q.subsIncl, fillIncl = q.fillIncl, strokeIncl = q.strokeIncl;
		const uint32_t myTasks = q.myTasks, myTask0 = q.myTask0, type = q.type, na = q.na, recIndex = q.recIndex, serV = q.serV, serS = q.serS, serM = q.serM, serF = q.serF;
		const float ptx = q.ptx, pty = q.pty, tessTol = q.tessTol;
		const uint64_t d = q.d, ownerBase = q.ownerBase;
		const float* pa = ps.args + q.argOff;
		const float* mtx = A.draws[valid ? d : 0].mtx;
			// ---- place --------------------------------------------------------------------------------------
			const long long gl = (long long)baseV + runV + (long long)excl; // output index of my first vertex
			uint32_t limit = (valid && !serialDraw) ? (uint32_t)(rawCnt < 0 ? 0 : rawCnt) : 0u;
			if (endNearFirst && limit > 0 && spTotal > 2) { --limit; } // my last vertex is the one pathClose removes
			float* out = A.poly + 2 * gl;
			if (cubicMask) {
				if (staged) {
					// per task: place relative to the chunk's first vertex (+ 64: a pathClose pop makes excl -1 at most), limit, owner lane
					if (isCubic) {
						uint32_t pl = (uint32_t)(excl + 64), left = deep ? 0u : limit;
						const uint32_t own = (uint32_t)lane << 24; // a task has at most 2^10 leaves: the limit leaves room for the owner's lane
						for (uint32_t j = 0; j < myTasks; ++j) {
							const uint32_t kj = L.tinfo[myTask0 + j].x & 0xFFFFFFu;
							const uint32_t lj = left < kj ? left : kj;
							L.tinfo[myTask0 + j] = make_uint2(pl, lj | own);
							pl += kj; left -= lj;
						}
					}
					__syncthreads();
					float2* obase = (float2*)A.poly + ((long long)baseV + runV - 64);
					const float m0 = mtx[0], m1 = mtx[1], m2 = mtx[2], m3 = mtx[3], m4 = mtx[4], m5 = mtx[5];
					for (uint32_t l0 = 0; l0 < listN; l0 += 2 * VGX_WAVE) { // two entries per lane and step: their LDS chains run side by side
						const uint32_t liA = l0 + (uint32_t)lane, liB = liA + VGX_WAVE;
						const bool lvA = liA < listN, lvB = liB < listN;
						const float2 qA = L.list[lvA ? liA : (uint32_t)CAP], qB = L.list[lvB ? liB : (uint32_t)CAP];
						const uint32_t tgA = lvA ? (uint32_t)L.tag[liA] : 0u, tgB = lvB ? (uint32_t)L.tag[liB] : 0u;
						const uint2 tiA = L.tinfo[tgA & 63u], tiB = L.tinfo[tgB & 63u];
						float ax = qA.x, ay = qA.y, bx = qB.x, by = qB.y;
						if (XFORM) { // the owner's transform comes through shuffles (all lanes take part)
							const int oA = (int)(tiA.y >> 24), oB = (int)(tiB.y >> 24);
							const float a0 = __shfl(m0, oA), a1 = __shfl(m1, oA), a2 = __shfl(m2, oA), a3 = __shfl(m3, oA), a4 = __shfl(m4, oA), a5 = __shfl(m5, oA);
							const float b0 = __shfl(m0, oB), b1 = __shfl(m1, oB), b2 = __shfl(m2, oB), b3 = __shfl(m3, oB), b4 = __shfl(m4, oB), b5 = __shfl(m5, oB);
							const float nax = a0 * ax + a2 * ay + a4, nay = a1 * ax + a3 * ay + a5; // transformPos2D, vg_util.h:24-28
							const float nbx = b0 * bx + b2 * by + b4, nby = b1 * bx + b3 * by + b5;
							ax = nax; ay = nay; bx = nbx; by = nby;
						}
						if (lvA && (tgA >> 6) < (tiA.y & 0xFFFFFFu)) { obase[tiA.x + (tgA >> 6)] = make_float2(ax, ay); }
						if (lvB && (tgB >> 6) < (tiB.y & 0xFFFFFFu)) { obase[tiB.x + (tgB >> 6)] = make_float2(bx, by); }
					}
				}
			}
			if (valid && !serialDraw) {
				if (type == VGX_CMD_MOVE_TO || type == VGX_CMD_LINE_TO) {
					if (limit > 0) {
						V2 p = v2(ptx, pty);
						if (XFORM) { p = v2xform(p, mtx); }
						*(float2*)out = make_float2(p.x, p.y);
					}
				} else if (isCubic && (!staged || deep)) {
					(void)0; // walked again below, 32 owners at a time
				} else if (type == VGX_CMD_POLYLINE && limit < (uint32_t)VGX_WAVE) {
					const uint32_t skip = (na >> 1) - (uint32_t)rawCnt;
					for (uint32_t i = 0; i < limit; ++i) {
						V2 p = v2(pa[2 * (i + skip)], pa[2 * (i + skip) + 1]);
						if (XFORM) { p = v2xform(p, mtx); }
						*(float2*)(out + 2 * i) = make_float2(p.x, p.y);
					}
				}
			}
			{ // cubics that are not in the list (the chunk overflowed it, or the cubic nests deeper than the LDS levels)
				uint64_t againMask = wave_ballot(isCubic && (!staged || deep));
				while (againMask) {
					const int ar = __popcll(againMask & lanemask_lt(lane));
					const bool me = isCubic && (!staged || deep) && ar < 32 && ((againMask >> lane) & 1ull);
					if (me) {
						const VgxCmdRec rr = ps.cmdrec[recIndex]; // (read again: rare path, and the record is not kept across the walk)
						float q1x = rr.a[0], q1y = rr.a[1], q2x = rr.a[2], q2y = rr.a[3], qex = rr.a[4], qey = rr.a[5];
						if (rr.type == VGX_CMD_QUAD_TO) {
							qex = rr.a[2]; qey = rr.a[3];
							vgx_quad_to_cubic(rr.start[0], rr.start[1], rr.a[0], rr.a[1], qex, qey, &q1x, &q1y, &q2x, &q2y);
						}
						LdsStack2<VGX_F1_LV> st2;
						st2.a = &L.stack[2 * ar]; st2.b = &L.stack[2 * ar + 1];
						FastCubicSink<true, XFORM> sink;
						sink.prev = v2(rr.start[0], rr.start[1]); sink.n = 0; sink.slow = false; sink.out = out; sink.writeLimit = limit; sink.mtx = mtx; sink.begin();
						vgx_flatten_cubic(rr.start[0], rr.start[1], q1x, q1y, q2x, q2y, qex, qey, tessTol, st2, sink);
						sink.flush();
					}
					uint64_t m = againMask; int dropped = 0;
					while (m && dropped < 32) { m &= m - 1; ++dropped; }
					againMask = m;
				}
			}
			{ // long POLYLINE commands: the wave moves them together (path.cpp:684-705 copies the points verbatim)
				uint64_t longMask = wave_ballot(valid && !serialDraw && type == VGX_CMD_POLYLINE && limit >= (uint32_t)VGX_WAVE);
				while (longMask) {
					const int src = __builtin_ctzll(longMask);
					longMask &= longMask - 1;
					const uint64_t gS = wave_bcast_u64((uint64_t)gl, src);
					const float* paS = (const float*)wave_bcast_u64((uint64_t)(pa + 2 * ((na >> 1) - (uint32_t)rawCnt)), src);
					const float* mS = (const float*)wave_bcast_u64((uint64_t)mtx, src);
					const uint32_t limS = (uint32_t)wave_bcast((int)limit, src);
					const float m0 = mS[0], m1 = mS[1], m2 = mS[2], m3 = mS[3], m4 = mS[4], m5 = mS[5];
					float2* outS = (float2*)A.poly + gS;
					for (uint32_t i = (uint32_t)lane; i < limS; i += VGX_WAVE) {
						const float2 q = *(const float2*)(paS + 2 * (size_t)i);
						outS[i] = XFORM ? make_float2(m0 * q.x + m2 * q.y + m4, m1 * q.x + m3 * q.y + m5) : q;
					}
				}
			}
			// ---- sub-path records, per-draw records ---------------------------------------------------------------
			const uint64_t subsGlobalIncl = baseS + runS + (uint64_t)subInclC;   // sub-paths up to and including my lane
			const uint64_t meshGlobalIncl = baseM + runM + (uint64_t)meshInclC;
			if (lastInSub) {
				vgx_subpath r;
				r.first_vertex = (uint64_t)(gl - (long long)spBefore);
				r.num_vertices = (uint32_t)spTotal;
				r.flags = closedHere ? 1u : 0u;
				// my sub-path's number: the sub-paths that exist up to my lane end with mine (its head is at or before my lane)
				A.subs[subsGlobalIncl - 1] = r;
			}
			if (X.has_empty && valid && drawHead) {
				// draws of EMPTY paths in front of my draw (no lane ever sees them): their records hold the running totals,
				// as the scan over draws of the two-phase entry gives them
				vgx_draw_info de;
				de.first_poly_vertex = (uint64_t)gl; de.first_subpath = subsGlobalIncl - (uint64_t)subAll; de.first_mesh = meshGlobalIncl - (uint64_t)meshAll;
				de.num_poly_vertices = 0; de.num_subpaths = 0; de.num_meshes = 0; de.flags = 0;
				for (uint64_t e = d; e > 0 && A.cmd_prefix[e - 1] == ownerBase; --e) { A.dinfo[e - 1] = de; }
			}
			if (valid && drawLast) {
				vgx_draw_info di;
				if (serialDraw) {
					di.num_poly_vertices = serV; di.num_subpaths = serS; di.num_meshes = serM; di.flags = serF; // counts from k_flatten_serial<count>; places from here
					di.first_poly_vertex = (uint64_t)gl;
					di.first_subpath = subsGlobalIncl - serS;
					di.first_mesh = meshGlobalIncl - serM;
					A.dinfo[d] = di;
				} else if (!slowDraw) {
					di.first_poly_vertex = (uint64_t)(gl - (long long)inDrawBefore);
					di.first_subpath = subsGlobalIncl - (uint64_t)subsIncl;
					di.first_mesh = meshGlobalIncl - (uint64_t)(fillIncl + strokeIncl);
					di.num_poly_vertices = (uint32_t)(inDrawBefore + cnt);
					di.num_subpaths = (uint32_t)subsIncl;
					di.num_meshes = (uint32_t)(fillIncl + strokeIncl);
					di.flags = ((uint32_t)fillIncl << 1);
					A.dinfo[d] = di;
				}
			}
			const uint64_t serialTails = wave_ballot(serialTail);
			if (serialTails && lane == 0) { atomicAdd(&T->flat_serial_draws, (unsigned long long)__popcll(serialTails)); }
	};

	for (;;) {
		// ---- next segment ---------------------------------------------------------------------------------------------
		const unsigned long long c0 = F1_CLK();
		const unsigned long long w0 = F1_WALL();
		while (N.stage < 4) { // what the last segment's wait did not cover (all of it for the wave's first segment)
			const F1NextRaw NR = f1_next_issue(N, A, X, T, lane, numSegments);
			f1_next_commit(N, NR, numSegments);
		}
		const unsigned long long t = N.t;
		if (t >= numSegments) { F1_FLUSH(); return; }
		F1_DBG(t, 0, w0);
		if (f1_fatal(__hip_atomic_load(&T->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { F1_FLUSH(); return; } // an error anywhere ends the call (nobody waits for us: they see it too)
		const uint64_t d0 = N.d0, d1 = N.d1, C0 = N.C0, C1 = N.C1;
		const DrawWindow W0 = N.W;
		N.stage = 0; // the next front starts in this segment's look-back
		const bool single = C1 - C0 <= (uint64_t)VGX_WAVE;
		const bool lastOfGroup = (t & 63ull) == 63ull || t + 1 == numSegments; // this wave publishes the group's records
		long long totV = 0; uint64_t totS = 0, totM = 0;   // the segment's totals
		uint64_t baseV = 0, baseS = 0, baseM = 0;          // sums of the segments in front
		bool published = false, writeOk = true;

		for (int pass = single ? 1 : 0; pass < 2; ++pass) {
			const bool stage = pass == 1;
			long long runV = 0; uint64_t runS = 0, runM = 0;  // placed in front of the current chunk, inside the segment
			uint64_t dcur = d0;
			int carryDrawVerts = 0, carrySpVerts = 0, carrySubs = 0, carryFill = 0, carryStroke = 0, carrySlow = 0, carrySpExists = 0;
			wbase = d0; W = W0; // the window over the segment's first draws came with the segment's front

			for (uint64_t chunk = C0; chunk < C1 || (chunk == C0 && !published && pass == 1); chunk += VGX_WAVE) {
				const uint64_t ci = chunk + lane;
				const bool valid = ci < C1;
				// ---- decode my command instance (as k_flatten, vgx_flatten.hip) -----------------------------------------
				uint64_t d = d0;
				uint32_t type = VGX_CMD_CLOSE, cflags = 0, na = 0;
				bool drawHead = false, drawLast = false, serialDraw = false;
				float scale = 1.0f, tol = 0.25f;
				uint32_t fillFlags = 0, strokeFlags = 0;
				const vgx_draw* dr = A.draws;
				const uint64_t lastKey = chunk + (VGX_WAVE - 1);
				if (!(wave_bcast_u64(W.prefix, VGX_WAVE - 1) > lastKey) || dcur < wbase) {
					wbase = dcur;
					W = draw_window_load(A, wbase, lane);
				}
				const bool windowCovers = wave_bcast_u64(W.prefix, VGX_WAVE - 1) > lastKey;
				const uint32_t wrel = window_rel(W.prefix, chunk);
				const int ownerOfs = window_owner_rel(wrel, valid ? (uint32_t)lane : 0u);
				const uint32_t orel = (uint32_t)__shfl((int)wrel, ownerOfs);
				const int firstOwner = __popcll(wave_ballot(W.prefix <= chunk)) - 1;
				uint64_t ownerBase = orel > 0 ? chunk + orel : wave_bcast_u64(W.prefix, firstOwner < 0 ? 0 : firstOwner);
				uint32_t pc0 = (uint32_t)__shfl((int)(W.pc0 | ((W.serial & 1u) << 31)), ownerOfs);
				uint32_t serialStatic = pc0 >> 31;
				pc0 &= 0x7FFFFFFFu;
				VgxCmdRec rec;
				rec.type = VGX_CMD_CLOSE; rec.flags = 0; rec.na = 0; rec.arg_off = 0; rec.start[0] = 0.0f; rec.start[1] = 0.0f;
				for (int i = 0; i < 8; ++i) { rec.a[i] = 0.0f; }
				uint32_t serV = 0, serS = 0, serM = 0, serF = 0; // a serial draw's counts (k_flatten_serial<count>), at its last command
				uint32_t recIndex = 0;
				if (valid) {
					if (windowCovers) {
						d = wbase + (uint64_t)ownerOfs;
					} else { // more than 63 draws begin inside this chunk (1-command or empty paths)
						d = find_owner_u64(A.cmd_prefix, d0, d1, ci);
						ownerBase = A.cmd_prefix[d];
						const uint32_t path = A.draws[d].path;
						pc0 = ps.path_cmd_begin[path];
						serialStatic = ps.path_flags[path] & VGX_PF_SERIAL;
					}
					dr = A.draws + d;
					const uint32_t k = (uint32_t)(ci - ownerBase);
					recIndex = pc0 + k;
					rec = ps.cmdrec[recIndex];
					type = rec.type; cflags = rec.flags; na = rec.na;
					drawHead = (k == 0);
					drawLast = (cflags & VGX_CF_LAST_IN_PATH) != 0;
					scale = dr->scale; tol = dr->tess_tol;
					fillFlags = dr->fill_flags; strokeFlags = dr->stroke_flags;
					serialDraw = serialStatic != 0;
					if (readFlags) {
						const uint32_t fl = A.dinfo[d].flags;
						serialDraw = serialDraw || (fl & 1u) != 0;
						if (serialDraw && drawLast) { // counted by k_flatten_serial: the draw is one opaque block here
							const vgx_draw_info si = A.dinfo[d];
							serV = si.num_poly_vertices; serS = si.num_subpaths; serM = si.num_meshes; serF = si.flags;
						}
					}
				}
				const float* a = rec.a;
				const float* pa = ps.args + rec.arg_off;
				const V2 start = v2(rec.start[0], rec.start[1]);
				// what the phases behind the walk need of the command record, so that the record itself does not stay in registers
				// across the walk (the kernel must fit two waves per SIMD): the point of a MOVE_TO / LINE_TO, and the two epsilon
				// tests of pathClose (path.cpp:716-722) -- my end point against the sub-path's first point (a[6..7] of every record)
				const float ptx = a[0], pty = a[1];
				bool endNearFirst = false, startNearFirst = false, lineSlow = false;
				if (valid) {
					const V2 firstPt = v2(rec.a[6], rec.a[7]);
					const V2 endp = (type == VGX_CMD_POLYLINE) ? (na >= 2 ? v2(pa[na - 2], pa[na - 1]) : v2(0.0f, 0.0f)) : (type == VGX_CMD_CUBIC_TO ? v2(a[4], a[5]) : (type == VGX_CMD_QUAD_TO ? v2(a[2], a[3]) : v2(a[0], a[1])));
					endNearFirst = (cflags & VGX_CF_NEXT_IS_CLOSE) != 0 && v2near(endp, firstPt);
					startNearFirst = v2near(start, firstPt);
					lineSlow = v2near(start, v2(a[0], a[1]));
				}

				// ---- tasks: the chunk's cubics, compacted; non-flat roots cut in two while lanes are free ----------------
				float c1x = a[0], c1y = a[1], c2x = a[2], c2y = a[3], ex = a[4], ey = a[5];
				const bool isCubic = valid && !serialDraw && (type == VGX_CMD_CUBIC_TO || type == VGX_CMD_QUAD_TO);
				if (isCubic && type == VGX_CMD_QUAD_TO) {
					ex = a[2]; ey = a[3];
					vgx_quad_to_cubic(start.x, start.y, a[0], a[1], ex, ey, &c1x, &c1y, &c2x, &c2y);
				}
				const float tessTol = tol / (scale * scale);
				const uint64_t cubicMask = wave_ballot(isCubic);
				const unsigned long long c1 = F1_CLK();
				if (chunk == C0) { F1_DBG(t, 3, F1_WALL() | ((unsigned long long)blockIdx.x << 40)); }
				F1_ACC(0, c1 - (chunk == C0 && (pass == 1) == single ? c0 : c1)); // ticket + segment table + draw window + command records (first chunk of a segment)
				F1_ACC(6, 1ull);
				int cnt = 0;
				bool slow = false, exists = false, closedHere = false, deep = false;
				uint32_t listN = 0, myTask0 = 0, myTasks = 0;
				if (cubicMask) { // wave-uniform
					// ---- task records: every cubic starts as one task; while at most half of the lanes hold a task, every task whose
					// node is not flat is cut in two (path.cpp:105-129, the walk's own step) -- 32 cubics become 64 tasks in one round,
					// a lone 145-segment cubic 64 tasks in six. The records live in the (still empty) leaf list, two buffers.
					float* prmA = (float*)L.list;                       // [F1_PARAM_WORDS][64]
					float* prmB = prmA + F1_PARAM_WORDS * VGX_WAVE;
					int numTasks = __popcll(cubicMask);
					if (isCubic) {
						const int r = __popcll(cubicMask & lanemask_lt(lane));
						prmA[0 * 64 + r] = start.x; prmA[1 * 64 + r] = start.y; prmA[2 * 64 + r] = c1x; prmA[3 * 64 + r] = c1y;
						prmA[4 * 64 + r] = c2x; prmA[5 * 64 + r] = c2y; prmA[6 * 64 + r] = ex; prmA[7 * 64 + r] = ey;
						prmA[8 * 64 + r] = tessTol; prmA[9 * 64 + r] = __int_as_float(lane);
					}
					__syncthreads(); // one-wave workgroup: LDS wait only
					v2f Q1, Q2, Q3, Q4; float qtol = 1.0f; int qown = 0;
					Q1.x = 0.0f; Q1.y = 0.0f; Q2 = Q1; Q3 = Q1; Q4 = Q1;
					for (;;) {
						const bool ta = lane < numTasks;
						if (ta) {
							Q1.x = prmA[0 * 64 + lane]; Q1.y = prmA[1 * 64 + lane]; Q2.x = prmA[2 * 64 + lane]; Q2.y = prmA[3 * 64 + lane];
							Q3.x = prmA[4 * 64 + lane]; Q3.y = prmA[5 * 64 + lane]; Q4.x = prmA[6 * 64 + lane]; Q4.y = prmA[7 * 64 + lane];
							qtol = prmA[8 * 64 + lane]; qown = __float_as_int(prmA[9 * 64 + lane]);
						}
						if (numTasks > VGX_WAVE / 2) { break; }
						const v2f rd = Q4 - Q1;
						const v2f ra2 = Q2 - Q4, ra3 = Q3 - Q4;
						const v2f rsw = rd.yx;
						const v2f rm2 = ra2 * rsw, rm3 = ra3 * rsw;
						const float rd2 = __builtin_fabsf(rm2.x - rm2.y), rd3 = __builtin_fabsf(rm3.x - rm3.y);
						const float rd23 = rd2 + rd3;
						const v2f rdd = rd * rd;
						const bool nf = ta && !(rd23 * rd23 <= qtol * (rdd.x + rdd.y));
						const uint64_t nfMask = wave_ballot(nf);
						if (!nfMask) { break; }
						const v2f R12 = (Q1 + Q2) * 0.5f, R23 = (Q2 + Q3) * 0.5f, R34 = (Q3 + Q4) * 0.5f;
						const v2f R123 = (R12 + R23) * 0.5f, R234 = (R23 + R34) * 0.5f;
						const v2f R1234 = (R123 + R234) * 0.5f;
						const int ni = lane + __popcll(nfMask & lanemask_lt(lane));
						if (ta) {
							const float ownf = __int_as_float(qown);
							if (nf) {
								prmB[0 * 64 + ni] = Q1.x; prmB[1 * 64 + ni] = Q1.y; prmB[2 * 64 + ni] = R12.x; prmB[3 * 64 + ni] = R12.y;
								prmB[4 * 64 + ni] = R123.x; prmB[5 * 64 + ni] = R123.y; prmB[6 * 64 + ni] = R1234.x; prmB[7 * 64 + ni] = R1234.y;
								prmB[8 * 64 + ni] = qtol; prmB[9 * 64 + ni] = ownf;
								prmB[0 * 64 + ni + 1] = R1234.x; prmB[1 * 64 + ni + 1] = R1234.y; prmB[2 * 64 + ni + 1] = R234.x; prmB[3 * 64 + ni + 1] = R234.y;
								prmB[4 * 64 + ni + 1] = R34.x; prmB[5 * 64 + ni + 1] = R34.y; prmB[6 * 64 + ni + 1] = Q4.x; prmB[7 * 64 + ni + 1] = Q4.y;
								prmB[8 * 64 + ni + 1] = qtol; prmB[9 * 64 + ni + 1] = ownf;
							} else {
								prmB[0 * 64 + ni] = Q1.x; prmB[1 * 64 + ni] = Q1.y; prmB[2 * 64 + ni] = Q2.x; prmB[3 * 64 + ni] = Q2.y;
								prmB[4 * 64 + ni] = Q3.x; prmB[5 * 64 + ni] = Q3.y; prmB[6 * 64 + ni] = Q4.x; prmB[7 * 64 + ni] = Q4.y;
								prmB[8 * 64 + ni] = qtol; prmB[9 * 64 + ni] = ownf;
							}
						}
						numTasks += __popcll(nfMask);
						__syncthreads();
						float* sw = prmA; prmA = prmB; prmB = sw;
					}
					const bool taskActive = lane < numTasks;
					// first task of every owner -> tinfo[owner].y (tasks are in owner order)
					{
						const uint32_t prevOwn = wave_from_prev_u32((uint32_t)qown, 0xFFFFFFFFu);
						if (taskActive && (lane == 0 || prevOwn != (uint32_t)qown)) { L.tinfo[qown].y = (uint32_t)lane; }
					}
					__syncthreads(); // the records are read: the list may take leaves
					uint32_t kT = 0; bool slowT = false, abortT = false;
#ifdef VGX_F1_PROFILE
					listN = f1_task_walk<CAP>(L, lane, taskActive, Q1, Q2, Q3, Q4, qtol, stage, &kT, &slowT, &abortT, &prof5);
#else
					listN = f1_task_walk<CAP>(L, lane, taskActive, Q1, Q2, Q3, Q4, qtol, stage, &kT, &slowT, &abortT);
#endif
					L.tinfo[lane].x = kT | (slowT ? F1_TF_SLOW : 0u) | (abortT ? F1_TF_ABORT : 0u);
					__syncthreads();
					if (isCubic) {
						const uint64_t later = cubicMask & ~lanemask_le(lane);
						myTask0 = L.tinfo[lane].y;
						const uint32_t t1 = later ? L.tinfo[__builtin_ctzll(later)].y : (uint32_t)numTasks;
						myTasks = t1 - myTask0;
						uint32_t sum = 0, fl = 0;
						for (uint32_t j = 0; j < myTasks; ++j) { const uint32_t w = L.tinfo[myTask0 + j].x; sum += w & 0xFFFFFFu; fl |= w; }
						cnt = (int)sum;
						slow = (fl & F1_TF_SLOW) != 0;
						deep = (fl & F1_TF_ABORT) != 0;
					}
					// cubics a task gave up on: counted by the owner with the full-depth walk, 32 owners at a time (two stack columns each)
					uint64_t deepMask = wave_ballot(deep);
					while (deepMask) { // wave-uniform
						const int dr_ = __popcll(deepMask & lanemask_lt(lane));
						const bool mine = deep && dr_ < 32 && ((deepMask >> lane) & 1ull);
						if (mine) {
							LdsStack2<VGX_F1_LV> st2;
							st2.a = &L.stack[2 * dr_]; st2.b = &L.stack[2 * dr_ + 1];
							FastCubicSink<false, false> sink;
							sink.prev = start; sink.n = 0; sink.slow = false;
							vgx_flatten_cubic(start.x, start.y, c1x, c1y, c2x, c2y, ex, ey, tessTol, st2, sink);
							cnt = (int)sink.n;
							slow = sink.slow;
						}
						// drop the 32 lowest set bits
						uint64_t m = deepMask; int dropped = 0;
						while (m && dropped < 32) { m &= m - 1; ++dropped; }
						deepMask = m;
					}
				}
				const bool staged = stage && listN <= (uint32_t)CAP; // wave-uniform: the list holds every leaf of the chunk
#ifdef VGX_F1_PROFILE
				if (g_f1_dbg && t < g_f1_dbg_n) { // flags of the timeline record: a cubic took the full-depth redo / the list overflowed
					const unsigned long long fl = (wave_ballot(deep) ? (1ull << 63) : 0ull) | ((stage && !staged) ? (1ull << 62) : 0ull);
					if (lane == 0 && fl) { atomicOr(&g_f1_dbg[t * 4 + 3], fl); }
					if (lane == 0 && stage) { atomicOr(&g_f1_dbg[t * 4 + 3], (unsigned long long)(listN & 0xFFFu) << 50); } // leaves of the chunk (12 bits)
				}
#endif
				const unsigned long long c2 = F1_CLK();
				F1_ACC(1, c2 - c1); // root step + task records + walk + counts back
				if (valid && !serialDraw) {
					switch (type) {
					case VGX_CMD_MOVE_TO: cnt = 1; exists = true; break;
					case VGX_CMD_LINE_TO: cnt = 1; slow = lineSlow; break;
					case VGX_CMD_POLYLINE: {
						const uint32_t npts = na >> 1;
						cnt = (int)npts - ((npts > 0 && v2near(start, v2(pa[0], pa[1]))) ? 1 : 0);
						slow = cnt == 0;
					} break;
					default: break;
					}
				}
				const int rawCnt = cnt;

				// ---- segmented bookkeeping (as the count pass of k_flatten) ------------------------------------------------
				const uint64_t drawHeads = wave_ballot(valid && drawHead);
				const uint64_t subHeads = wave_ballot(valid && (cflags & VGX_CF_STARTS_SUB));
				const int dh = seg_head(drawHeads, lane);
				const int sh = seg_head(subHeads, lane);
				{
					const int incl1 = wave_incl_scan(cnt, lane);
					const int spBefore1 = seg_rel(incl1 - cnt, sh, carrySpVerts);
					if (valid && !serialDraw && type == VGX_CMD_CLOSE && spBefore1 > 2) { // pathClose, path.cpp:707-726
						closedHere = true;
						if (startNearFirst) { cnt = -1; }
					}
				}
				const bool serialTail = valid && serialDraw && drawLast; // carries the whole serial draw's counts
				const int cntAll = serialTail ? (int)serV : cnt;
				const int incl = wave_incl_scan(cntAll, lane);
				const int excl = incl - cntAll;
				const int inDrawBefore = seg_rel(excl, dh, carryDrawVerts);
				const int spBefore = seg_rel(excl, sh, carrySpVerts);
				const int spTotal = spBefore + cnt;
				const uint64_t existMask = wave_ballot(valid && exists);
				const uint64_t mine = seg_mask_upto(dh, lane);
				const int subsIncl = __popcll(existMask & mine) + (dh < 0 ? carrySubs : 0);
				const int headExists = (sh < 0) ? carrySpExists : (int)((existMask >> sh) & 1ull);
				const bool lastInSub = valid && !serialDraw && (cflags & VGX_CF_LAST_IN_SUB) && headExists;
				const bool fillHere = lastInSub && (fillFlags & VGX_FILL_ENABLE) && spTotal >= 3;
				const bool strokeHere = lastInSub && (strokeFlags & VGX_STROKE_ENABLE) && spTotal >= 2;
				const uint64_t fillMask = wave_ballot(fillHere);
				const uint64_t strokeMask = wave_ballot(strokeHere);
				const int fillIncl = __popcll(fillMask & mine) + (dh < 0 ? carryFill : 0);
				const int strokeIncl = __popcll(strokeMask & mine) + (dh < 0 ? carryStroke : 0);
				const uint64_t slowMask = wave_ballot(valid && slow);
				const bool slowDraw = ((slowMask & mine) != 0) || (dh < 0 && carrySlow);
				// sub-paths / meshes in front of me inside the chunk (serial draws count as blocks at their last command)
				const int subAll = serialTail ? (int)serS : (exists ? 1 : 0);
				const int meshAll = serialTail ? (int)serM : ((fillHere ? 1 : 0) + (strokeHere ? 1 : 0));
				const int subInclC = wave_incl_scan(subAll, lane);
				const int meshInclC = wave_incl_scan(meshAll, lane);

				const int nvalid = (int)((C1 - chunk) < (uint64_t)VGX_WAVE ? (C1 - chunk) : (uint64_t)VGX_WAVE);
				const int LL = nvalid > 0 ? nvalid - 1 : 0;
				const int chunkV = nvalid > 0 ? wave_bcast(incl, LL) : 0;
				const int chunkS = nvalid > 0 ? wave_bcast(subInclC, LL) : 0;
				const int chunkM = nvalid > 0 ? wave_bcast(meshInclC, LL) : 0;

				// degenerate draws found here: flagged in dinfo and listed for k_flatten_serial; the kernel runs again (pass 1)
				{
					const bool newSlow = valid && drawLast && !serialDraw && slowDraw;
					const uint64_t sm = wave_ballot(newSlow);
					if (sm && stage) {
						unsigned long long sbase = 0;
						if (lane == 0) { sbase = atomicAdd(&T->num_serial_list, (unsigned long long)__popcll(sm)); T->flat_redo = 1u; }
						sbase = wave_bcast_u64(sbase, 0);
						if (newSlow) {
							A.serial_list[sbase + (uint64_t)__popcll(sm & lanemask_lt(lane))] = (uint32_t)d;
							vgx_draw_info di;
							di.first_poly_vertex = 0; di.first_subpath = 0; di.first_mesh = 0; di.num_poly_vertices = 0; di.num_subpaths = 0; di.num_meshes = 0; di.flags = 1u;
							A.dinfo[d] = di;
						}
					}
				}

				const unsigned long long c3 = F1_CLK();
				F1_ACC(2, c3 - c2); // bookkeeping
				if (!stage) {
					totV += chunkV; totS += (uint64_t)chunkS; totM += (uint64_t)chunkM;
				} else {
					if (single) { totV = chunkV; totS = (uint64_t)chunkS; totM = (uint64_t)chunkM; }
					if (!published) {
						// ---- the segment's place in the output: publish, look back, publish ----------------------------
						// (a segment's net vertex count is never negative: a pathClose pops a vertex of its own sub-path)
						const uint64_t myV = (uint64_t)(totV < 0 ? 0 : totV);
						if (lane == 0) { f1_publish(X.segs, t, F1_A, myV, totS, totM); }
						F1_DBG(t, 1, F1_WALL());
#ifdef VGX_F1_PROFILE
						if (!f1_lookback(X.segs, X.grps, t, lastOfGroup, myV, totS, totM, lane, T, &baseV, &baseS, &baseM, &prof7, &prof8)) { F1_FLUSH(); return; }
#else
						if (!f1_lookback(X.segs, X.grps, t, lastOfGroup, myV, totS, totM, lane, T, &baseV, &baseS, &baseM)) { F1_FLUSH(); return; }
#endif
						F1_DBG(t, 2, F1_WALL());
						const uint64_t endV = baseV + myV, endS = baseS + totS, endM = baseM + totM;
						if (endS >= F1_SM_LIMIT || endM >= F1_SM_LIMIT) { // the records carry 31-bit sub-path / mesh sums: split the batch
							if (lane == 0) { atomicCAS(&T->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_RANGE); }
							F1_FLUSH();
							return;
						}
						if (endV > X.cap_poly || endS > X.cap_subs) {
							writeOk = false;
							if (lane == 0) { atomicCAS(&T->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_NOSPACE); }
						}
						published = true;
					}
					const unsigned long long c4 = F1_CLK();
					F1_ACC(3, c4 - c3); // publish + look-back
					if (writeOk && nvalid > 0) {
						F1Pend q;
						q.flags = (valid ? 1u : 0u) | (serialDraw ? 2u : 0u) | (endNearFirst ? 4u : 0u) | (isCubic ? 8u : 0u) | (deep ? 16u : 0u) | (lastInSub ? 32u : 0u) | (closedHere ? 64u : 0u)
							| (drawHead ? 128u : 0u) | (drawLast ? 256u : 0u) | (slowDraw ? 512u : 0u) | (serialTail ? 1024u : 0u);
						q.excl = excl; q.rawCnt = rawCnt; q.spTotal = spTotal; q.myTasks = myTasks; q.myTask0 = myTask0; q.type = type; q.ptx = ptx; q.pty = pty; q.na = na;
						q.argOff = rec.arg_off; q.recIndex = recIndex; q.tessTol = tessTol; q.spBefore = spBefore; q.subInclC = subInclC; q.meshInclC = meshInclC; q.subAll = subAll; q.meshAll = meshAll;
						q.serV = serV; q.serS = serS; q.serM = serM; q.serF = serF; q.inDrawBefore = inDrawBefore; q.cnt = cnt; q.subsIncl = subsIncl; q.fillIncl = fillIncl; q.strokeIncl = strokeIncl;
						q.d = d; q.ownerBase = ownerBase;
						placeChunk(q, cubicMask, staged, listN, baseV, baseS, baseM, runV, runS, runM);
					}
					runV += chunkV; runS += (uint64_t)chunkS; runM += (uint64_t)chunkM;
					F1_ACC(4, F1_CLK() - c4); // placement + records
				}

				// ---- carries into the next chunk (taken from the last valid lane) ---------------------------------------------
				if (nvalid > 0) {
					const int lastIsDrawLast = wave_bcast((int)drawLast, LL);
					const int lastIsSubLast = wave_bcast((int)((cflags & VGX_CF_LAST_IN_SUB) != 0), LL);
					const int nDraw = wave_bcast(inDrawBefore + cntAll, LL);
					const int nSp = wave_bcast(spTotal, LL);
					const int nSubs = wave_bcast(subsIncl, LL);
					const int nFill = wave_bcast(fillIncl, LL);
					const int nStroke = wave_bcast(strokeIncl, LL);
					const int nSlow = wave_bcast((int)slowDraw, LL);
					const int nHeadExists = wave_bcast(headExists, LL);
					carryDrawVerts = lastIsDrawLast ? 0 : nDraw;
					carrySubs = lastIsDrawLast ? 0 : nSubs;
					carryFill = lastIsDrawLast ? 0 : nFill;
					carryStroke = lastIsDrawLast ? 0 : nStroke;
					carrySlow = lastIsDrawLast ? 0 : nSlow;
					carrySpVerts = (lastIsDrawLast || lastIsSubLast) ? 0 : nSp;
					carrySpExists = (lastIsDrawLast || lastIsSubLast) ? 0 : nHeadExists;
					dcur = wave_bcast_u64(d, LL);
				}
				__syncthreads(); // the list / task table are reused by the next chunk
			}
			if (pass == 0) {
				// several chunks: every chunk is counted, the segment can take its place
				const uint64_t myV = (uint64_t)(totV < 0 ? 0 : totV);
				if (lane == 0) { f1_publish(X.segs, t, F1_A, myV, totS, totM); }
				#ifdef VGX_F1_PROFILE
						if (!f1_lookback(X.segs, X.grps, t, lastOfGroup, myV, totS, totM, lane, T, &baseV, &baseS, &baseM, &prof7, &prof8)) { F1_FLUSH(); return; }
#else
						if (!f1_lookback(X.segs, X.grps, t, lastOfGroup, myV, totS, totM, lane, T, &baseV, &baseS, &baseM)) { F1_FLUSH(); return; }
#endif
				const uint64_t endV = baseV + myV, endS = baseS + totS, endM = baseM + totM;
				if (endS >= F1_SM_LIMIT || endM >= F1_SM_LIMIT) {
					if (lane == 0) { atomicCAS(&T->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_RANGE); }
					F1_FLUSH();
					return;
				}
				if (endV > X.cap_poly || endS > X.cap_subs) {
					writeOk = false;
					if (lane == 0) { atomicCAS(&T->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_NOSPACE); }
				}
				published = true;
			}
		}
	}
}

// Segment table: seg_draw[k] = first draw whose first command instance is >= k * segItems (lower bound in cmd_prefix), for
// k in [0, numSegments]; one thread per draw writes the entries that fall into its command range. Also the per-draw
// records of draws WITHOUT commands (no lane of k_flat1 ever sees them) and the batch's command-instance total.
__global__ __launch_bounds__(256) void k_f1_seg_table(VgxFlattenArgs A, VgxF1Args X)
{
	const uint64_t n = A.ndraws;
	const uint64_t totalCmds = A.cmd_prefix[n];
	const uint64_t S = vgx_f1_segment_items(totalCmds, n, X.seg_max);
	const uint64_t numSegments = (totalCmds + S - 1) / S;
	{ // the look-back records of this batch's segments start empty (the buffer is sized by a host-side bound, not cleared whole)
		unsigned long long* w = (unsigned long long*)X.segs;
		const uint64_t nw = numSegments * (sizeof(VgxF1Seg) / 8);
		for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (uint64_t)gridDim.x * blockDim.x) { w[i] = 0ull; }
		unsigned long long* wg = (unsigned long long*)X.grps;
		const uint64_t ng = ((numSegments + 63) / 64) * (sizeof(VgxF1Seg) / 8);
		for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += (uint64_t)gridDim.x * blockDim.x) { wg[i] = 0ull; }
	}
	for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d <= n; d += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t p = A.cmd_prefix[d];
		const uint64_t lo = d == 0 ? 0 : A.cmd_prefix[d - 1] / S + 1;
		uint64_t hi = p / S;
		if (d == n) { hi = numSegments; } // entries past the last draw's first command
		for (uint64_t k = lo; k <= hi && k <= numSegments; ++k) { X.seg_draw[k] = d; }
	}
}

// Between the two runs of k_flat1: only when the first run found degenerate draws. Clears the look-back records and the ticket.
__global__ __launch_bounds__(256) void k_f1_redo_clear(VgxFlattenArgs A, VgxF1Args X)
{
	VgxTotals* T = A.totals;
	if (T->flat_redo == 0u) { return; }
	const uint64_t totalCmds = A.cmd_prefix[A.ndraws];
	const uint64_t S = vgx_f1_segment_items(totalCmds, A.ndraws, X.seg_max);
	const uint64_t numSegments = (totalCmds + S - 1) / S;
	unsigned long long* w = (unsigned long long*)X.segs;
	const uint64_t nw = numSegments * (sizeof(VgxF1Seg) / 8);
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (uint64_t)gridDim.x * blockDim.x) { w[i] = 0ull; }
	unsigned long long* wg = (unsigned long long*)X.grps;
	const uint64_t ng = ((numSegments + 63) / 64) * (sizeof(VgxF1Seg) / 8);
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += (uint64_t)gridDim.x * blockDim.x) { wg[i] = 0ull; }
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		T->flat_ticket = 0ull; T->flat_serial_draws = 0ull;
		// the first run sized the degenerate draws by their nominal counts (>= the exact ones): a capacity verdict from it does not stand
		if (T->status == VGX_E_NOSPACE) { T->status = VGX_OK; }
	}
}

// The exact serial count of the draws the first run listed (build_mode-style list walk, counts only, no heap).
__global__ __launch_bounds__(256) void k_f1_serial_count_list(VgxFlattenArgs A)
{
	VgxTotals* T = A.totals;
	if (T->flat_redo == 0u || (T->status != VGX_OK && T->status != VGX_E_NOSPACE)) { return; }
	const VgxPathSetDev& ps = A.ps;
	PrivStackF1 stack;
	const uint64_t nwork = (uint64_t)T->num_serial_list;
	for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwork; w += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t d = (uint64_t)A.serial_list[w];
		const vgx_draw* dr = A.draws + d;
		const uint32_t path = dr->path;
		const uint32_t pc0 = ps.path_cmd_begin[path], pc1 = ps.path_cmd_begin[path + 1];
		PathSim<false, false> sim;
		sim.scale = dr->scale; sim.tol = dr->tess_tol; sim.mtx = dr->mtx; sim.poly = A.poly;
		sim.drawIndex = (uint32_t)d; sim.fillFlags = dr->fill_flags; sim.strokeFlags = dr->stroke_flags; sim.draw = dr;
		sim.polyBase = 0; sim.subs = nullptr; sim.subBase = 0; sim.mdesc = nullptr; sim.mprep = nullptr; sim.mtab = nullptr; sim.meshBase = 0;
		sim.numFillTotal = 0; sim.limit = 0;
		sim.init();
		sim.run(ps, pc0, pc1, stack);
		vgx_draw_info di;
		di.first_poly_vertex = 0; di.first_subpath = 0; di.first_mesh = 0;
		di.num_poly_vertices = sim.nverts; di.num_subpaths = sim.nsubs; di.num_meshes = sim.nfill + sim.nstroke;
		di.flags = 1u | (sim.nfill << 1);
		A.dinfo[d] = di;
	}
}

// Totals and status for the caller: the last segment's inclusive prefix.
__global__ __launch_bounds__(256) void k_f1_publish(VgxFlattenArgs A, VgxF1Args X, vgx_sizes* devSizes, uint32_t* devStatus)
{
	VgxTotals* T = A.totals;
	const uint64_t totalCmds = A.cmd_prefix[A.ndraws];
	const uint64_t S = vgx_f1_segment_items(totalCmds, A.ndraws, X.seg_max);
	const uint64_t numSegments = (totalCmds + S - 1) / S;
	uint32_t st = T->status;
	vgx_sizes z = T->sizes; // num_cmd_instances from the scan
	z.num_poly_vertices = 0; z.num_subpaths = 0; z.num_meshes = 0; z.num_serial_draws = 0; z.num_vertices = 0; z.num_indices = 0;
	if (!f1_fatal(st) && numSegments > 0) {
		const VgxF1Seg g = X.grps[(numSegments - 1) / 64]; // the last group's inclusive prefix
		if ((g.w[0] >> 62) != F1_P || (g.w[1] >> 62) != F1_P) { st = VGX_E_INTERNAL; }
		z.num_poly_vertices = g.w[0] & F1_VAL; z.num_subpaths = (g.w[1] & F1_VAL) >> 31; z.num_meshes = g.w[1] & (F1_SM_LIMIT - 1);
		z.num_serial_draws = T->flat_serial_draws;
	}
	if (threadIdx.x == 0) {
		T->sizes = z;
		T->status = st;
		T->flat_tag = X.tag;
		if (devSizes) { *devSizes = z; }
		if (devStatus) { *devStatus = st; }
	}
	if (X.has_empty && st == VGX_OK) { // draws of empty paths behind the batch's last command: no segment holds them
		vgx_draw_info de;
		de.first_poly_vertex = z.num_poly_vertices; de.first_subpath = z.num_subpaths; de.first_mesh = z.num_meshes;
		de.num_poly_vertices = 0; de.num_subpaths = 0; de.num_meshes = 0; de.flags = 0;
		for (uint64_t d = X.seg_draw[numSegments] + threadIdx.x; d < A.ndraws; d += blockDim.x) { A.dinfo[d] = de; }
	}
}

} // namespace

void vgx_launch_flat1(const VgxFlattenArgs& a, const VgxF1Args& x, int waves, int cap, bool hasStaticSerial, hipStream_t s)
{
	VgxF1Args x0 = x; x0.pass = 0; x0.read_flags = hasStaticSerial ? 1 : 0;
	VgxF1Args x1 = x; x1.pass = 1; x1.read_flags = 1;
	hipLaunchKernelGGL(k_f1_seg_table, dim3(1024), dim3(256), 0, s, a, x0);
	auto launch = [&](const VgxF1Args& xx, bool redo) {
#define F1_LAUNCH(C, XF) do { if (redo) { hipLaunchKernelGGL((k_flat1<C, XF, true>), dim3(waves), dim3(VGX_WAVE), 0, s, a, xx); } \
	else { hipLaunchKernelGGL((k_flat1<C, XF, false>), dim3(waves), dim3(VGX_WAVE), 0, s, a, xx); } } while (0)
		if (a.apply_transform) {
			if (cap >= 3072) { F1_LAUNCH(3072, true); } else if (cap >= 2048) { F1_LAUNCH(2048, true); } else if (cap >= 1664) { F1_LAUNCH(1664, true); } else { F1_LAUNCH(1024, true); }
		} else {
			if (cap >= 3072) { F1_LAUNCH(3072, false); } else if (cap >= 2048) { F1_LAUNCH(2048, false); } else if (cap >= 1664) { F1_LAUNCH(1664, false); } else { F1_LAUNCH(1024, false); }
		}
#undef F1_LAUNCH
	};
	launch(x0, false);
	// degenerate draws found by the first run (none, normally: these three exit at once)
	hipLaunchKernelGGL(k_f1_serial_count_list, dim3(64), dim3(256), 0, s, a);
	hipLaunchKernelGGL(k_f1_redo_clear, dim3(256), dim3(256), 0, s, a, x1);
	launch(x1, true);
}

#ifdef VGX_F1_PROFILE
extern "C" int vgx_f1_debug_occupancy(int cap)
{
	int n = -1;
	hipError_t e;
	if (cap >= 3072) { e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_flat1<3072, true, false>, VGX_WAVE, 0); }
	else if (cap >= 2048) { e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_flat1<2048, true, false>, VGX_WAVE, 0); }
	else if (cap >= 1664) { e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_flat1<1664, true, false>, VGX_WAVE, 0); }
	else { e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_flat1<1024, true, false>, VGX_WAVE, 0); }
	return e == hipSuccess ? n : -(int)e;
}
extern "C" int vgx_f1_debug_buffer(void* p, unsigned long long n)
{
	unsigned long long* q = (unsigned long long*)p;
	if (hipMemcpyToSymbol(HIP_SYMBOL(g_f1_dbg), &q, sizeof(q)) != hipSuccess) { return 1; }
	if (hipMemcpyToSymbol(HIP_SYMBOL(g_f1_dbg_n), &n, sizeof(n)) != hipSuccess) { return 1; }
	return 0;
}
#endif

void vgx_launch_flat1_publish(const VgxFlattenArgs& a, const VgxF1Args& x, vgx_sizes* devSizes, uint32_t* devStatus, hipStream_t s)
{
	hipLaunchKernelGGL(k_f1_publish, dim3(1), dim3(256), 0, s, a, x, devSizes, devStatus);
}
