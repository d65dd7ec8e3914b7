// vgx_inst.hip -- k_flatten_inst: single-pass flatten of INSTANCED batches, one lane per instance.
//
// vgx_tessellate's flatten stage for batches whose draws repeat the same sequence of P paths (draws[i].path ==
// draws[i mod P].path: a drawing submitted for many instances -- the shape vgx_draw was designed around, 64 B of input per
// path instance). k_flatten_build maps one LANE to one path COMMAND; neighbouring lanes then hold different commands
// and the adaptive subdivision of pathCubicTo (path.cpp:86-182) runs in lock-step at 31 % lane use (26 % of the
// cubics of the headline drawing are one segment, 2.5 % need >= 17 steps). Here the 64 lanes of a wave are 64 INSTANCES
// of the same path: every lane executes the path's commands in order with its own draw record (transform, scale,
// tolerance, flags), i.e. vg::Path's own sequential algorithm (InstCore, vgx_inst.h), and the walk diverges only as far
// as the instances' tolerances differ. What falls away with the command-parallel mapping: the segmented scans and
// carries, the leaf slots in LDS and their copy loop, the owner windows, the exactness escape hatch (epsilon
// de-duplication and pathClose are evaluated in order, so no draw is ever handed to k_flatten_serial as "degenerate").
//   - command records come through the SCALAR cache (one s_load_dwordx16 per command per wave: the path is wave-uniform);
//   - a lane appends its vertices to a lane-private block of the polyline heap (InstCore::grow); the wave's lanes take
//     their blocks with ONE atomic per round of allocations;
//   - sub-path records, per-draw counts and the list of statically serial draws (arcs / closed shapes) are written in
//     the format k_flatten_build writes, so everything downstream (scan over draws, k_flatten_gather, k_fill, k_stroke)
//     is unchanged and the meshes are bit-identical.
// The periodic structure is found by vgx_tessellate_count (k_inst_find / k_inst_verify below) and re-checked on the
// device by every vgx_tessellate (OpCmdPrefix, vgx_scan_ops.h): when the draw records were rewritten in a way that
// breaks it, this kernel exits at once and k_flatten_build does the batch.
#include "vgx_internal.h"
#include "vgx_wave.h"
#include "vgx_walk.h"
#include "vgx_inst.h"
#include "vgx_scan.h"

namespace {

#ifndef VGX_INST_LDS_LEVELS
#define VGX_INST_LDS_LEVELS 4
#endif
#ifndef VGX_INST_STAGE
#define VGX_INST_STAGE 8 /* vertices a lane parks in LDS before it writes them as one aligned piece (8: 64 bytes, 4: 32 bytes) */
#endif

// loads of data no kernel of the sequence writes after upload / after the preceding scan: constant address space, so
// that wave-uniform addresses become scalar loads
template<class T> __device__ __forceinline__ const __attribute__((address_space(4))) T* as_const(const T* p)
{
	return (const __attribute__((address_space(4))) T*)(uintptr_t)p;
}

struct InstEnvDev
{
	float* poly;
	uint64_t cap;
	uint32_t lb;
	unsigned long long* cursor;
	uint32_t* status;
	int lane;
	float2* stage; // s_stage: 8 staged vertices per lane, slot j of lane l at [j][(l + 4 j) mod 64] (the rotation keeps the
	               // cooperative flush, which reads four slots of sixteen lanes at once, off the same banks)
	__device__ __forceinline__ float2* slotOf(uint32_t j, uint32_t l) const { return stage + j * VGX_WAVE + ((l + 4u * j) & (VGX_WAVE - 1u)); }
	__device__ __forceinline__ float2* slot(uint32_t j) const { return slotOf(j, (uint32_t)lane); }

	// Vertex stores. A lane's vertices go to consecutive heap addresses; written one by one, 64 lanes x 8 bytes land in 64
	// different cache lines per store instruction and every line stays partly written for several commands (measured:
	// 1.15 of the kernel's 2.1 ms). So a vertex is parked in LDS at slot (address / 8) mod 8 and a lane writes a whole
	// aligned 64-byte piece (four 16-byte stores back to back) when it has filled the last slot. Blocks start on piece
	// boundaries (InstCore::grow), a moved sub-path is re-emitted through here, and a vertex pathClose pops leaves its
	// slot behind unchanged, so the slots below the write position always hold the current piece's vertices.
	__device__ __forceinline__ void emit(float* wp, float x, float y)
	{
#ifdef VGX_EXP_INST_NOEMIT
		asm volatile("" :: "v"(x), "v"(y), "v"(wp));
		return;
#endif
		const uint32_t slot = ((uint32_t)(uintptr_t)wp >> 3) & (VGX_INST_STAGE - 1u);
		*this->slot(slot) = make_float2(x, y);
		if (slot == VGX_INST_STAGE - 1u) {
			float4* dst = (float4*)(wp - 2 * (VGX_INST_STAGE - 1));
			float2 v[VGX_INST_STAGE];
#pragma unroll
			for (int i = 0; i < VGX_INST_STAGE - 1; ++i) { v[i] = *this->slot((uint32_t)i); }
			v[VGX_INST_STAGE - 1] = make_float2(x, y);
#ifndef VGX_EXP_INST_NOSTORE
#pragma unroll
			for (int j = 0; j < VGX_INST_STAGE / 2; ++j) { dst[j] = make_float4(v[2 * j].x, v[2 * j].y, v[2 * j + 1].x, v[2 * j + 1].y); }
#endif
		}
	}
	// emit() of the lock-step walk (all lanes of `act` arrive together). When the whole wave fills its last slot in the
	// same step -- instances of one scale, whose write positions advance together -- the pieces leave TRANSPOSED: store i
	// covers the lanes' pieces 16 i .. 16 i + 15, four lanes per 64-byte piece, so that a store instruction presents 16
	// whole pieces to the memory pipeline instead of 64 quarter pieces. The owner lane's address comes through a shuffle.
	__device__ __forceinline__ void emitLockstep(float* wp, float x, float y, uint64_t act)
	{
#if VGX_INST_STAGE == 8 && !defined(VGX_INST_NO_COOP_FLUSH) && !defined(VGX_EXP_INST_NOEMIT)
		const uint32_t slotIdx = ((uint32_t)(uintptr_t)wp >> 3) & 7u;
		if (act == ~0ull && __ballot(slotIdx == 7u) == ~0ull) {
			*this->slot(7u) = make_float2(x, y);
			const uint64_t piece = (uint64_t)(uintptr_t)(wp - 14);
			const uint32_t q = (uint32_t)lane & 3u;
#pragma unroll
			for (uint32_t i = 0; i < 4; ++i) {
				const uint32_t s = 16u * i + ((uint32_t)lane >> 2);
				const float2 a = *slotOf(2u * q, s), b = *slotOf(2u * q + 1u, s);
				const uint64_t dst = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)piece, (int)s) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(piece >> 32), (int)s) << 32)) + 16u * q;
#ifndef VGX_EXP_INST_NOSTORE
				typedef float f4v __attribute__((ext_vector_type(4)));
				f4v v; v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
				*(__attribute__((address_space(1))) f4v*)dst = v; // a shuffled address is an integer: global space, explicitly
#endif
			}
			return;
		}
#endif
		(void)act;
		emit(wp, x, y);
	}
	// the staged part of the current piece (slots below the write position) goes to the heap: before a sub-path is
	// moved (its tail is read back from the heap) and when the wave is done
	__device__ __forceinline__ void flushPartial(float* wp)
	{
		const uint32_t n = ((uint32_t)(uintptr_t)wp >> 3) & (VGX_INST_STAGE - 1u);
		float2* dst = (float2*)(wp - 2 * n);
		for (uint32_t i = 0; i < n; ++i) { dst[i] = *slot(i); }
	}
	// The write position went BACK from wpOld to wpNew (a cubic is redone): if that left the piece wpOld was in, the slots
	// have been reused by later pieces; the piece wpNew is in was written out completely meanwhile, so its vertices below
	// wpNew come back from the heap.
	__device__ __forceinline__ void rewind(float* wpNew, float* wpOld)
	{
		if (((uintptr_t)wpNew / (8 * VGX_INST_STAGE)) == ((uintptr_t)wpOld / (8 * VGX_INST_STAGE))) { return; }
		__threadfence_block();
		const uint32_t n = ((uint32_t)(uintptr_t)wpNew >> 3) & (VGX_INST_STAGE - 1u);
		const float2* src = (const float2*)(wpNew - 2 * n);
		for (uint32_t i = 0; i < n; ++i) { *slot(i) = src[i]; }
	}
	__device__ __forceinline__ void flushForMove(float* wp)
	{
		flushPartial(wp);
		__threadfence_block(); // the copy loop reads what this lane wrote
	}
	// Called by whichever lanes ran out of room (any subset of the wave): one atomic for all of them.
	__device__ __forceinline__ bool alloc(uint64_t want, uint64_t* base)
	{
		const uint64_t mask = __ballot(1);
		const uint32_t wlo = (uint32_t)want, whi = (uint32_t)(want >> 32);
		uint64_t total = 0, off = 0;
		for (uint64_t m = mask; m != 0; m &= m - 1) {
			const int l = __builtin_ctzll(m);
			const uint64_t w = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)whi, l) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wlo, l);
			if (lane == l) { off = total; }
			total += w;
		}
		const int leader = __builtin_ctzll(mask);
		unsigned long long b = 0;
		if (lane == leader) { b = atomicAdd(cursor, (unsigned long long)total); }
		b = wave_bcast_u64(b, leader);
		if (b + total > cap || b + total < b) {
			if (lane == leader) { atomicCAS(status, (uint32_t)VGX_OK, (uint32_t)VGX_E_NOSPACE); }
			return false;
		}
		*base = b + off;
		return true;
	}
};

typedef InstCore<InstEnvDev> InstLane;

// -DVGX_INST_PROFILE: wave clock (100 MHz) summed over all waves per phase of k_flatten_inst, read back through
// vgx_get_failure_info().prof: 0 task prologue, 1 cubic walks, 2 command loops (cubic walks included), 3 whole waves,
// 4 tasks, 5 cubics
#ifdef VGX_INST_PROFILE
#define IPROF_T(var) const uint64_t var = wall_clock64()
#define IPROF_ACC(acc, t0, t1) acc += (t1) - (t0)
#else
#define IPROF_T(var) do {} while (0)
#define IPROF_ACC(acc, t0, t1) do {} while (0)
#endif

// value of lane `l` (wave-uniform l): v_readlane_b32, whatever the exec mask
__device__ __forceinline__ uint32_t rl_u32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ float rl_f32(float v, uint32_t l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)l)); }

// Lock-step walk. The lanes of a wave hold the same cubic, and as long as they also take the same decisions (instances of
// one scale: always) the walk needs no per-lane control flow: one ballot per node, wave-uniform (scalar) branches, the
// pending count and the stack position in scalar registers, no midpoints on leaf nodes. Returns true when the cubic was
// finished that way. On the first node where the lanes disagree -- and on anything rare: a vertex inside the epsilon
// ball, a full heap block, more than LV pending halves -- it returns false: the caller takes the lane back to the state
// before the cubic and walks it with the per-lane loop (keeping the two loops' variables apart is what keeps this one
// free of register copies; a hand-over in the middle of the cubic cost ~25 v_mov per node).
template<int LV>
__device__ __forceinline__ bool inst_cubic_lockstep(InstLane& L, v2f P2, v2f P3, v2f P4, float2* stackLane)
{
	v2f P1;
	P1.x = L.last.x; P1.y = L.last.y;
	v2f prev = P1;
	const float tessTol = L.tessTol;
	int pendU = 0;
	uint32_t spU = 0;
	const uint64_t act = __ballot(1);
	// One latch, one exit, two wave-uniform `if`s that update the node in place: with early returns / continues the
	// compiler turned the loop into a state machine that copied every loop variable twice per node (~25 v_mov).
	bool cont, ok;
	do {
		const v2f d = P4 - P1;
		const v2f a2 = P2 - P4, a3 = P3 - P4;
		const v2f dsw = d.yx;
		const v2f m2 = a2 * dsw, m3 = a3 * dsw;
		const float d2 = __builtin_fabsf(m2.x - m2.y), d3 = __builtin_fabsf(m3.x - m3.y);
		const float d23 = d2 + d3;
		const v2f dd = d * d;
		const bool flat = d23 * d23 <= tessTol * (dd.x + dd.y);
		const v2f e = prev - P4; // pathAddVertex: lastVertex - (x, y)
		const v2f ee = e * e;
		const uint64_t fm = __ballot(flat);
		const uint64_t rare = __ballot((ee.x + ee.y < VGM_EPSILON) || L.room == 0);
		const bool desc = fm == 0 && pendU < LV;        // every lane descends into the left half
		const bool leaf = fm == act && rare == 0;       // every lane has a flat piece and stores its end point
		const bool pop = leaf && pendU > 0;
		if (desc) {
			const v2f P12 = (P1 + P2) * 0.5f, P23 = (P2 + P3) * 0.5f, P34 = (P3 + P4) * 0.5f;
			const v2f P123 = (P12 + P23) * 0.5f, P234 = (P23 + P34) * 0.5f;
			const v2f P1234 = (P123 + P234) * 0.5f;
			stackLane[spU] = make_float2(P234.x, P234.y);
			stackLane[spU + VGX_WAVE] = make_float2(P34.x, P34.y);
			stackLane[spU + 2 * VGX_WAVE] = make_float2(P4.x, P4.y);
			spU += 3 * VGX_WAVE;
			++pendU;
			P2 = P12; P3 = P123; P4 = P1234;
		}
		if (leaf) {
			const float ox = L.m0 * P4.x + L.m2 * P4.y + L.m4, oy = L.m1 * P4.x + L.m3 * P4.y + L.m5; // transformPos2D, vg_util.h:24-28
			if (__ballot(L.spN < 3u) != 0) { L.note(ox, oy); }
			L.env.emitLockstep(L.wp, ox, oy, act);
			L.wp += 2;
			--L.room;
			++L.spN;
			prev = P4;
			P1 = P4;
			if (pop) {
				spU -= 3 * VGX_WAVE;
				--pendU;
				const float2 q2 = stackLane[spU], q3 = stackLane[spU + VGX_WAVE], q4 = stackLane[spU + 2 * VGX_WAVE];
				P2.x = q2.x; P2.y = q2.y; P3.x = q3.x; P3.y = q3.y; P4.x = q4.x; P4.y = q4.y;
			}
		}
		cont = desc || pop;
		ok = desc || leaf;
	} while (cont);
	if (!ok) { return false; }
	L.last = v2(prev.x, prev.y);
	return true;
}

// pathCubicTo for one lane of the instanced kernel: the hand-shaped walk of build_flatten_hot (vgx_walk.h: packed float
// pairs, single exit, pending right halves in LDS as [level][3][64] float2) with pathAddVertex inlined at the leaves --
// epsilon test against the last STORED vertex (path.cpp:769-775), transform, store to the lane's heap block. Arithmetic and
// its order are those of path.cpp:107-170. Returns false when the cubic nests deeper than LV pending halves; the caller
// restores the lane and redoes the cubic with the full-depth stack (vertices already written are simply overwritten).
template<int LV>
__device__ __forceinline__ bool inst_cubic_hot(InstLane& L, v2f P2, v2f P3, v2f P4, float2* stackLane)
{
	v2f P1;
	P1.x = L.last.x; P1.y = L.last.y;
	v2f prev = P1;
	const float tessTol = L.tessTol;
	int pending = 0;
	bool more = true, aborted = false;
	uint32_t sp = 0;
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
				const v2f e = prev - P4; // pathAddVertex: lastVertex - (x, y)
				const v2f ee = e * e;
				if (!(ee.x + ee.y < VGM_EPSILON)) {
					if (L.room == 0) { L.grow(); }
					float2 o;
					o.x = L.m0 * P4.x + L.m2 * P4.y + L.m4; // transformPos2D, vg_util.h:24-28
					o.y = L.m1 * P4.x + L.m3 * P4.y + L.m5;
					if (L.spN < 3u) { L.note(o.x, o.y); }
					L.env.emit(L.wp, o.x, o.y);
					L.wp += 2;
					--L.room;
					++L.spN;
					prev = P4;
				}
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
	L.last = v2(prev.x, prev.y);
	return !aborted;
}

// Back to the state before the cubic: the write position follows the vertex count (a block switch that happened
// meanwhile stays), the staged piece is re-primed (InstEnvDev::rewind); vertices already written are simply overwritten.
__device__ __forceinline__ void inst_cubic_back(InstLane& L, uint32_t n0, V2 last0)
{
	const uint32_t back = L.spN - n0;
	if (back != 0 && !L.dead) {
		float* const now = L.wp;
		L.wp -= 2 * (uint64_t)back; L.room += back;
		L.env.rewind(L.wp, now);
	}
	L.spN = n0;
	L.last = last0;
}

// stackLane = &s_stack[lane], handed down from the kernel as an expression on the __shared__ array (NOT through stack.base:
// `stack` lives in private memory because of its deep[] levels, and a pointer loaded from there is a flat pointer -- the
// hot loop's pushes and pops would become flat_store / flat_load instead of ds_write / ds_read).
__device__ __forceinline__ void inst_cubic(InstLane& L, float c1x, float c1y, float c2x, float c2y, float x, float y, LdsStackT<VGX_INST_LDS_LEVELS>& stack, float2* stackLane)
{
	const uint32_t n0 = L.spN;
	const V2 last0 = L.last;
	v2f P2, P3, P4;
	P2.x = c1x; P2.y = c1y; P3.x = c2x; P3.y = c2y; P4.x = x; P4.y = y;
#ifndef VGX_INST_NO_LOCKSTEP
	if (inst_cubic_lockstep<VGX_INST_LDS_LEVELS>(L, P2, P3, P4, stackLane)) { return; }
	inst_cubic_back(L, n0, last0);
#endif
	if (!inst_cubic_hot<VGX_INST_LDS_LEVELS>(L, P2, P3, P4, stackLane)) {
		// deeper than the LDS levels: the full-depth walk from the cubic's root
		inst_cubic_back(L, n0, last0);
		L.cubicTo(c1x, c1y, c2x, c2y, x, y, stack);
	}
}

#ifdef VGX_INST_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(VGX_INST_WAVES_PER_EU)))
#endif
__global__ __launch_bounds__(VGX_WAVE) void k_flatten_inst(VgxFlattenArgs A)
{
	__shared__ float2 s_stack[VGX_INST_LDS_LEVELS * 3 * VGX_WAVE];
	__shared__ float2 s_stage[VGX_INST_STAGE * VGX_WAVE];
	const int lane = threadIdx.x;
	const bool grouped = A.inst_order != nullptr; // draws grouped by path through inst_order (else: periodic batch, closed form)
	if (A.totals->status != VGX_OK || (!grouped && A.totals->inst_mismatch != 0)) { return; }
	LdsStackT<VGX_INST_LDS_LEVELS> stack;
	stack.base = &s_stack[lane];

	const VgxPathSetDev& ps = A.ps;
	const uint32_t P = grouped ? 1u : A.inst_period;
	const uint64_t ninst = A.ndraws / P;
	const auto cprefix = as_const(A.cmd_prefix); // periodic: of the first instance = command offsets inside every instance
	const uint64_t C = grouped ? 1ull : cprefix[P]; // commands per instance
	const uint64_t G = (ninst + VGX_WAVE - 1) / VGX_WAVE;
	if (C == 0) { return; }
	// periodic: task t = instance group t / P, path slot t % P; grouped: task t = 64 draws of path inst_task_path[t]
	const uint64_t numTasks = grouped ? (uint64_t)A.totals->inst_num_tasks : G * P;

	uint64_t profPro = 0, profCubic = 0, profLoop = 0, profTasks = 0, profCubics = 0;
	(void)profPro; (void)profCubic; (void)profLoop; (void)profTasks; (void)profCubics;
	IPROF_T(tWave0);
	InstLane L;
	L.env.poly = A.poly; L.env.cap = A.caps.poly_vertices; L.env.lb = A.inst_block; L.env.cursor = &A.totals->poly_heap_cursor;
	L.env.status = &A.totals->status; L.env.lane = lane; L.env.stage = s_stage;
	L.initLane();

#ifdef VGX_INST_STATIC
	// contiguous share of the task list per wave (paths differ a lot in cost: the slowest wave decides)
	const uint64_t share = (numTasks + gridDim.x - 1) / gridDim.x;
	uint64_t t = (uint64_t)blockIdx.x * share;
	const uint64_t tEnd = (t + share < numTasks) ? t + share : numTasks;
	for (; t < tEnd; ++t) {
#else
	// Tasks are handed out one at a time through ticket counters (paths differ a lot in cost; a static share per wave left
	// the average wave idle for a third of the kernel). ONE counter does not keep up: 41 776 device-scope atomics on one
	// address take 0.5 ms (measured with the command loop compiled out), half of the kernel. So the task list is cut
	// into VGX_INST_POOLS contiguous pools with a counter each (128 bytes apart); a wave starts in pool blockIdx mod
	// POOLS and moves on to the next pool when its pool is empty. The next ticket is requested before the current task
	// is processed.
	const uint64_t poolTasks = (numTasks + VGX_INST_POOLS - 1) / VGX_INST_POOLS;
	uint32_t pool = blockIdx.x % VGX_INST_POOLS, poolsLeft = VGX_INST_POOLS;
	unsigned long long ticket = 0;
	if (lane == 0) { ticket = atomicAdd(&A.totals->inst_ticket[pool * 16], 1ull); }
	for (;;) {
		const uint64_t tl = wave_bcast_u64(ticket, 0);
		const uint64_t t = (uint64_t)pool * poolTasks + tl;
		if (tl >= poolTasks || t >= numTasks) { // this pool is empty: try the next one
			if (--poolsLeft == 0) { break; }
			pool = (pool + 1) % VGX_INST_POOLS;
			if (lane == 0) { ticket = atomicAdd(&A.totals->inst_ticket[pool * 16], 1ull); }
			continue;
		}
		if (lane == 0) { ticket = atomicAdd(&A.totals->inst_ticket[pool * 16], 1ull); }
#endif
		IPROF_T(tTask0);
		uint32_t path;
		bool valid;
		uint64_t d, srecBase;
		if (!grouped) {
			const uint64_t g = t / P;
			const uint32_t pcur = (uint32_t)(t - g * P);
			path = as_const(A.draws)[pcur].path; // == draws[i * P + pcur].path for every instance i (verified)
			const uint64_t slot = g * VGX_WAVE + (uint64_t)lane;
			valid = slot < ninst;
			const uint64_t inst = (A.inst_perm != nullptr && valid) ? (uint64_t)A.inst_perm[slot] : slot;
			d = inst * P + pcur;
			srecBase = inst * as_const(A.sub_prefix)[P] + as_const(A.sub_prefix)[pcur]; // sub-paths in front of this draw (periodic: closed form)
		} else {
			path = as_const(A.inst_task_path)[t];
			const uint64_t first = as_const(A.inst_start)[path], end = as_const(A.inst_start)[path + 1];
			const uint64_t idx = first + (t - as_const(A.inst_task_start)[path]) * VGX_WAVE + (uint64_t)lane;
			valid = idx < end;
			d = valid ? (uint64_t)A.inst_order[idx] : 0ull;
			srecBase = A.sub_prefix[d];
		}
		const uint32_t pc0 = as_const(ps.path_cmd_begin)[path], pc1 = as_const(ps.path_cmd_begin)[path + 1];
		if (pc0 == pc1) { continue; } // nothing to build: the draw's (zeroed) record stands
		if (as_const(ps.path_flags)[path] & VGX_PF_SERIAL) {
			// arcs / closed shapes: the exact one-lane-per-draw builder (k_flatten_serial), as in k_flatten_build
			const uint64_t sm = wave_ballot(valid);
			unsigned long long sbase = 0;
			if (lane == 0) { sbase = atomicAdd(&A.totals->num_serial_list, (unsigned long long)__popcll(sm)); }
			sbase = wave_bcast_u64(sbase, 0);
			if (valid) { A.serial_list[sbase + (uint64_t)__popcll(sm & lanemask_lt(lane))] = (uint32_t)d; }
			continue;
		}
		// The path's command records: lane l fetches record k0 + l (one coalesced read per 64 commands), the command loop
		// takes record k from lane k - k0 with v_readlane -- no memory latency per command.
		const VgxCmdRec* recs = ps.cmdrec + pc0;
		const uint32_t ncmd = pc1 - pc0;
		const vgx_draw* dr = A.draws + d;
		VgxSubRec* srec = A.sub_rec + srecBase;
		if (valid) { L.beginDraw(dr->mtx, dr->scale, dr->tess_tol, dr->fill_flags, dr->stroke_flags); }
#ifdef VGX_EXP_INST_FLAT
		L.tessTol = 3.0e38f; // experiment: every cubic is one segment
#endif
#ifdef VGX_INST_PROFILE
		asm volatile("" :: "v"(L.m0), "v"(L.m5), "v"(L.tessTol)); // the draw record has arrived
#endif
		IPROF_T(tLoop0);
		IPROF_ACC(profPro, tTask0, tLoop0);
		++profTasks;
		for (uint32_t k0 = 0; k0 < ncmd; k0 += VGX_WAVE) {
			const uint32_t kc = (ncmd - k0 < VGX_WAVE) ? ncmd - k0 : VGX_WAVE;
			const VgxCmdRec* mine = recs + k0 + ((uint32_t)lane < kc ? (uint32_t)lane : kc - 1);
			const uint4 h = *(const uint4*)mine;               // type, flags, na, arg_off
			const float2 a01 = *(const float2*)&mine->a[0];   // byte 24
			const float4 a25 = *(const float4*)&mine->a[2];   // byte 32
			// wait for the records HERE: left to the first v_readlane, the wait sits inside the command loop, where it also
			// waits for the vertex stores of the previous command (vmcnt counts loads and stores in order)
			asm volatile("" :: "v"(h.x), "v"(h.y), "v"(h.z), "v"(h.w), "v"(a01.x), "v"(a01.y), "v"(a25.x), "v"(a25.y), "v"(a25.z), "v"(a25.w));
			// The broadcasts stay OUTSIDE `if (valid)`: every lane has to execute the loads above (a load that is only used
			// under the condition may be sunk into it, and v_readlane would then read lanes that never loaded).
			for (uint32_t kk = 0; kk < kc; ++kk) {
				const uint32_t type = rl_u32(h.x, kk), cflags = rl_u32(h.y, kk), na = rl_u32(h.z, kk), argOff = rl_u32(h.w, kk);
				const float a0f = rl_f32(a01.x, kk), a1f = rl_f32(a01.y, kk);
				const float a2f = rl_f32(a25.x, kk), a3f = rl_f32(a25.y, kk), a4f = rl_f32(a25.z, kk), a5f = rl_f32(a25.w, kk);
#ifdef VGX_EXP_INST_NOCMD
				asm volatile("" :: "s"(type), "s"(cflags), "s"(na), "s"(argOff), "s"(a0f), "s"(a1f), "s"(a2f), "s"(a3f), "s"(a4f), "s"(a5f));
				continue;
#endif
				if (valid) {
					switch (type) {
					case VGX_CMD_MOVE_TO: L.moveTo(a0f, a1f); break;
					case VGX_CMD_LINE_TO: L.lineTo(a0f, a1f); break;
					case VGX_CMD_CUBIC_TO: {
#ifdef VGX_EXP_INST_NOCUBIC
						break;
#endif
						IPROF_T(tc0);
						inst_cubic(L, a0f, a1f, a2f, a3f, a4f, a5f, stack, &s_stack[lane]);
						IPROF_T(tc1);
						IPROF_ACC(profCubic, tc0, tc1);
						++profCubics;
					} break;
					case VGX_CMD_QUAD_TO: {
						float c1x, c1y, c2x, c2y;
						vgx_quad_to_cubic(L.last.x, L.last.y, a0f, a1f, a2f, a3f, &c1x, &c1y, &c2x, &c2y);
						inst_cubic(L, c1x, c1y, c2x, c2y, a2f, a3f, stack, &s_stack[lane]);
					} break;
					case VGX_CMD_CLOSE: L.close(); break;
					case VGX_CMD_POLYLINE: {
						const auto pa = as_const(ps.args) + argOff;
						uint32_t n = na >> 1, i0 = 0;
						if (L.spN > 0 && n > 0 && v2near(L.last, v2(pa[0], pa[1]))) { i0 = 1; } // path.cpp:691-696
						for (uint32_t i = i0; i < n; ++i) {
							const V2 q = v2(pa[2 * i], pa[2 * i + 1]);
							if (L.spN == 0) { L.first = q; }
							L.put(q);
						}
					} break;
					default: break; // shapes / arcs only occur in statically serial paths
					}
#ifndef VGX_EXP_INST_NOSUB
					if (cflags & VGX_CF_LAST_IN_SUB) { L.endSub(srec + L.nsubs); } // dense: one 16-byte record per sub-path, in draw order
#endif
				}
			}
		}
		if (valid) { A.dinfo[d] = L.drawInfo(); }
		IPROF_T(tLoop1);
		IPROF_ACC(profLoop, tLoop0, tLoop1);
	}
	L.env.flushPartial(L.wp);
#ifdef VGX_INST_PROFILE
	{
		const uint64_t tWave1 = wall_clock64();
		if (lane == 0) {
			atomicAdd(&A.totals->prof[0], (unsigned long long)profPro); atomicAdd(&A.totals->prof[1], (unsigned long long)profCubic);
			atomicAdd(&A.totals->prof[2], (unsigned long long)profLoop); atomicAdd(&A.totals->prof[3], (unsigned long long)(tWave1 - tWave0));
			atomicAdd(&A.totals->prof[4], (unsigned long long)profTasks); atomicAdd(&A.totals->prof[5], (unsigned long long)profCubics);
		}
	}
#endif
}

// ---- finding the period (vgx_tessellate_count) ----------------------------------------------------------------
// P = distance to the first repetition of draws[0].path; then every draw is compared with its image in the first
// period. A drawing that uses one path twice gets a too small candidate and fails the check: command-parallel kernel.
__global__ __launch_bounds__(256) void k_inst_find(const vgx_draw* draws, uint64_t ndraws, VgxTotals* totals)
{
	const uint32_t p0 = draws[0].path;
	unsigned long long best = ~0ull;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < ndraws; i += (uint64_t)gridDim.x * blockDim.x) {
		if (draws[i].path == p0) { best = i; break; } // a thread's indices ascend
	}
	if (best != ~0ull) { atomicMax(&totals->inst_detect_inv, ~0ull - best); } // totals are zeroed: keep the minimum as a maximum
}

// Effective flattening tolerance of a draw (InstLane::beginDraw) as an ordered integer: the bit patterns of positive finite
// floats ascend with their values. Anything else (a zero scale, NaN) is class 0.
__device__ __forceinline__ uint32_t inst_tol_bits(const vgx_draw* d)
{
	const float t = d->tess_tol / (d->scale * d->scale);
	const uint32_t b = __float_as_uint(t);
	return (t > 0.0f && b < 0x7F800000u) ? b : 0u;
}

__global__ __launch_bounds__(256) void k_inst_verify(const vgx_draw* draws, uint64_t ndraws, VgxTotals* totals)
{
	const unsigned long long inv = totals->inst_detect_inv;
	const unsigned long long P = ~0ull - inv;
	if (inv == 0 || ndraws % P != 0) {
		if (blockIdx.x == 0 && threadIdx.x == 0) { totals->inst_detect_bad = 1u; }
		return;
	}
	bool bad = false, varies = false;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndraws; i += (uint64_t)gridDim.x * blockDim.x) {
		const vgx_draw* img = draws + i % P;
		bad = bad || (draws[i].path != img->path);
		varies = varies || (inst_tol_bits(draws + i) != inst_tol_bits(img));
	}
	if (bad) { totals->inst_detect_bad = 1u; }
	if (varies) { totals->inst_tol_varies = 1u; }
}

// Range of the draws' tolerances (grouped mode with tolerance classes; the count pass, to decide on them)
__global__ __launch_bounds__(256) void k_inst_tol_range(const vgx_draw* draws, uint64_t ndraws, VgxTotals* totals)
{
	__shared__ uint32_t s_lo, s_hi;
	if (threadIdx.x == 0) { s_lo = 0xFFFFFFFFu; s_hi = 0u; }
	__syncthreads();
	uint32_t lo = 0xFFFFFFFFu, hi = 0u;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndraws; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t b = inst_tol_bits(draws + i);
		lo = b < lo ? b : lo; hi = b > hi ? b : hi;
	}
	for (int o = 32; o > 0; o >>= 1) {
		const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, o), h2 = (uint32_t)__shfl_xor((int)hi, o);
		lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
	}
	if ((threadIdx.x & 63) == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
	__syncthreads();
	if (threadIdx.x == 0 && s_lo <= s_hi) { atomicMax(&totals->inst_tol_lo_inv, ~s_lo); atomicMax(&totals->inst_tol_hi, s_hi); }
}

// ---- grouped mode: the draws sorted by path ------------------------------------------------------------------------
// For batches that reuse paths without repeating one sequence (culled or shuffled instances, several drawings mixed): a
// counting sort by path id. The order inside a path's range is whatever the atomics give -- any lane may take any draw
// of its path; only the heap locality of the emit kernels' reads depends on it.
// Atomics on neighbouring counters serialise per cache line in the memory-side cache (2.2 M draws over 240 paths = 8 lines:
// the two passes took 2.1 ms with plain global atomics), so a workgroup counts its slice of the draws in LDS and touches
// the global counters once per used path. The LDS table is direct for path sets of up to VGX_INST_LDS_PATHS paths and a
// one-probe hash (slot = path mod table size, first comer owns the slot) beyond: draws of a slot's owner count in LDS,
// everything else goes to the global counter at once -- many draws on few paths stay in LDS, a million draws on a million
// paths are a million atomics on a million different addresses, which do not serialise.
//
// Tolerance classes. Lanes of a wave walk a cubic in lock-step only while they take the same flat / split decisions, i.e.
// while their draws have (nearly) the same tess_tol / scale^2. When the instances differ in scale the sort key becomes
// (path, class), class = the draw's tolerance quantised into `nc` steps over the batch's range [lo, hi] of bit patterns
// (k_inst_tol_range): a path's range in `order` is then ascending in tolerance, and the 64 consecutive entries a wave
// takes differ by about 64 / (instances of the path) of the range. Tasks stay (path, 64 entries) -- a task may straddle
// two classes, which costs a few per-lane cubics, not correctness: the order inside a path's range is free.
// Tiger x 9216 with scales 0.5 .. 3.5 (bench.py tiger10k_varied): flatten_build 7.9 ms in the periodic mapping, 1.6 ms +
// 0.33 ms of sorting with 64 .. 1024 classes (256 by default, VGX_INST_CLASSES).
#define VGX_INST_LDS_PATHS 4096
#define VGX_INST_GROUP_THREADS 1024
#define VGX_INST_GROUP_BLOCKS 256
#define VGX_INST_NO_KEY 0xFFFFFFFFu
__device__ __forceinline__ void inst_slice(uint64_t n, uint64_t* lo, uint64_t* hi)
{
	const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
	const uint64_t l = per * blockIdx.x;
	*lo = l < n ? l : n;
	*hi = l + per < n ? l + per : n;
}
struct InstKeys
{
	uint32_t npaths, nc, lo, shift; // nc = 1: the key is the path
	__device__ __forceinline__ uint32_t nkeys() const { return npaths * nc; }
	__device__ __forceinline__ uint32_t key(const vgx_draw* d) const
	{
		const uint32_t p = d->path;
		if (p >= npaths) { return VGX_INST_NO_KEY; } // an invalid path id was reported by the command scan
		if (nc == 1u) { return p; }
		const uint32_t b = inst_tol_bits(d);
		uint32_t c = b > lo ? (b - lo) >> shift : 0u;
		c = c < nc ? c : nc - 1u;
		return p * nc + c;
	}
};
__device__ __forceinline__ InstKeys inst_keys(uint32_t npaths, uint32_t nc, const VgxTotals* totals)
{
	InstKeys k;
	k.npaths = npaths; k.nc = nc; k.lo = 0; k.shift = 0;
	if (nc > 1u) {
		const uint32_t lo = ~totals->inst_tol_lo_inv, hi = totals->inst_tol_hi;
		k.lo = lo;
		const uint32_t span = hi > lo ? hi - lo : 0u;
		while ((span >> k.shift) >= nc) { ++k.shift; }
	}
	return k;
}
// true: key p counts in LDS slot *slot of this workgroup
__device__ __forceinline__ bool inst_slot(uint32_t* s_key, uint32_t p, uint32_t npaths, uint32_t* slot)
{
	if (npaths <= VGX_INST_LDS_PATHS) { *slot = p; return true; }
	const uint32_t sl = p & (VGX_INST_LDS_PATHS - 1);
	*slot = sl;
	const uint32_t old = atomicCAS(&s_key[sl], VGX_INST_NO_KEY, p);
	return old == VGX_INST_NO_KEY || old == p;
}

__global__ __launch_bounds__(VGX_INST_GROUP_THREADS) void k_inst_hist(const vgx_draw* draws, uint64_t ndraws, uint32_t npathsIn, uint32_t nc, const VgxTotals* totals, uint32_t* hist)
{
	const InstKeys K = inst_keys(npathsIn, nc, totals);
	const uint32_t npaths = K.nkeys(); // below: "path" = sort key
	__shared__ uint32_t s_cnt[VGX_INST_LDS_PATHS];
	__shared__ uint32_t s_key[VGX_INST_LDS_PATHS];
	uint64_t lo, hi;
	inst_slice(ndraws, &lo, &hi);
	const bool hashed = npaths > VGX_INST_LDS_PATHS;
	const uint32_t nslots = hashed ? (uint32_t)VGX_INST_LDS_PATHS : npaths;
	for (uint32_t p = threadIdx.x; p < nslots; p += blockDim.x) { s_cnt[p] = 0; s_key[p] = hashed ? VGX_INST_NO_KEY : p; }
	__syncthreads();
	for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
		const uint32_t p = K.key(draws + i);
		if (p == VGX_INST_NO_KEY) { continue; }
		uint32_t slot;
		if (inst_slot(s_key, p, npaths, &slot)) { atomicAdd(&s_cnt[slot], 1u); } else { atomicAdd(&hist[p], 1u); }
	}
	__syncthreads();
	for (uint32_t sl = threadIdx.x; sl < nslots; sl += blockDim.x) {
		const uint32_t c = s_cnt[sl];
		if (c) { atomicAdd(&hist[s_key[sl]], c); }
	}
}

// Exclusive scans of the histogram (draw ranges, task ranges), the task -> path table, the totals: a device scan over the
// paths (one workgroup up to 1024 paths, three passes beyond -- a path set may hold a million one-cubic paths).
// exclusive scan of the (path, class) histogram: first entry of every key in `order`
struct OpInstKeyStart
{
	const uint32_t* hist;
	uint64_t nkeys;
	uint64_t* keyStart;
	__device__ uint64_t size() const { return nkeys; }
	__device__ Sum3 load(uint64_t k) const { Sum3 r = sum3_zero(); r.a = hist[k]; return r; }
	__device__ void store(uint64_t k, Sum3 e) const { keyStart[k] = e.a; }
	__device__ void finish(Sum3 t) const { keyStart[nkeys] = t.a; }
};

struct OpInstPlan
{
	const uint32_t* hist;     // nc == 1: draws per path
	const uint64_t* keyStart; // nc > 1: scan of the (path, class) histogram
	uint32_t nc;
	uint32_t npaths;
	uint64_t* start;
	uint64_t* taskStart;
	uint32_t* taskPath;
	uint64_t capTasks;
	VgxTotals* totals;
	__device__ uint64_t size() const { return npaths; }
	__device__ Sum3 load(uint64_t p) const
	{
		Sum3 r = sum3_zero();
		const uint64_t cnt = count(p);
		r.a = cnt; r.b = (cnt + VGX_WAVE - 1) / VGX_WAVE; r.c = cnt ? 1ull : 0ull;
		return r;
	}
	__device__ uint64_t count(uint64_t p) const { return nc == 1u ? (uint64_t)hist[p] : keyStart[(p + 1) * nc] - keyStart[p * nc]; }
	__device__ void store(uint64_t p, Sum3 e) const
	{
		start[p] = e.a;
		taskStart[p] = e.b;
		if (taskPath) {
			const uint64_t nt = (count(p) + VGX_WAVE - 1) / VGX_WAVE;
			for (uint64_t j = 0; j < nt; ++j) { if (e.b + j < capTasks) { taskPath[e.b + j] = (uint32_t)p; } }
		}
	}
	__device__ void finish(Sum3 t) const
	{
		start[npaths] = t.a;
		taskStart[npaths] = t.b;
		totals->inst_num_tasks = t.b;
		totals->inst_distinct = t.c;
		if (taskPath && t.b > capTasks) { atomicCAS(&totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_NOSPACE); }
	}
};

__global__ __launch_bounds__(VGX_INST_GROUP_THREADS) void k_inst_scatter(const vgx_draw* draws, uint64_t ndraws, uint32_t npathsIn, uint32_t nc, const VgxTotals* totals, const uint64_t* start, uint32_t* cursor, uint32_t* order)
{
	const InstKeys K = inst_keys(npathsIn, nc, totals);
	const uint32_t npaths = K.nkeys(); // below: "path" = sort key, start[] = first entry of every key
	__shared__ uint32_t s_cnt[VGX_INST_LDS_PATHS];
	__shared__ uint32_t s_base[VGX_INST_LDS_PATHS];
	__shared__ uint32_t s_key[VGX_INST_LDS_PATHS];
	uint64_t lo, hi;
	inst_slice(ndraws, &lo, &hi);
	const bool hashed = npaths > VGX_INST_LDS_PATHS;
	const uint32_t nslots = hashed ? (uint32_t)VGX_INST_LDS_PATHS : npaths;
	// the slice's own histogram -> one reservation per used path in the path's range -> ranks inside the reservation
	// (draws whose path does not own its hash slot take their place with a global atomic in the first pass)
	for (uint32_t p = threadIdx.x; p < nslots; p += blockDim.x) { s_cnt[p] = 0; s_key[p] = hashed ? VGX_INST_NO_KEY : p; }
	__syncthreads();
	for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
		const uint32_t p = K.key(draws + i);
		if (p == VGX_INST_NO_KEY) { continue; }
		uint32_t slot;
		if (inst_slot(s_key, p, npaths, &slot)) { atomicAdd(&s_cnt[slot], 1u); }
		else { order[start[p] + atomicAdd(&cursor[p], 1u)] = (uint32_t)i; }
	}
	__syncthreads();
	for (uint32_t sl = threadIdx.x; sl < nslots; sl += blockDim.x) {
		const uint32_t c = s_cnt[sl];
		s_base[sl] = c ? atomicAdd(&cursor[s_key[sl]], c) : 0u;
		s_cnt[sl] = 0;
	}
	__syncthreads();
	for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
		const uint32_t p = K.key(draws + i);
		if (p == VGX_INST_NO_KEY) { continue; }
		const uint32_t sl = hashed ? (p & (VGX_INST_LDS_PATHS - 1)) : p;
		if (s_key[sl] == p) { order[start[p] + s_base[sl] + atomicAdd(&s_cnt[sl], 1u)] = (uint32_t)i; }
	}
}

} // namespace

void vgx_launch_inst_group(const vgx_draw* draws, uint64_t ndraws, uint32_t npaths, uint32_t nc, uint32_t* hist, uint32_t* cursor, uint64_t* keyStart, uint64_t* start,
	uint64_t* taskStart, uint32_t* taskPath, uint64_t capTasks, uint32_t* order, VgxTotals* totals, void* scanPartial, hipStream_t s)
{
	const uint64_t nkeys = (uint64_t)npaths * nc;
	(void)hipMemsetAsync(hist, 0, (nkeys + 1) * sizeof(uint32_t), s);
	if (nc > 1) { hipLaunchKernelGGL(k_inst_tol_range, dim3(512), dim3(256), 0, s, draws, ndraws, totals); }
	hipLaunchKernelGGL(k_inst_hist, dim3(VGX_INST_GROUP_BLOCKS), dim3(VGX_INST_GROUP_THREADS), 0, s, draws, ndraws, npaths, nc, (const VgxTotals*)totals, hist);
	if (nc > 1) {
		OpInstKeyStart ks;
		ks.hist = hist; ks.nkeys = nkeys; ks.keyStart = keyStart;
		vgx_device_scan(ks, (Sum3*)scanPartial, s, nkeys);
	}
	OpInstPlan op;
	op.hist = hist; op.keyStart = keyStart; op.nc = nc; op.npaths = npaths; op.start = start; op.taskStart = taskStart; op.taskPath = taskPath; op.capTasks = capTasks; op.totals = totals;
	vgx_device_scan(op, (Sum3*)scanPartial, s, npaths);
	if (order) {
		(void)hipMemsetAsync(cursor, 0, (nkeys + 1) * sizeof(uint32_t), s);
		hipLaunchKernelGGL(k_inst_scatter, dim3(VGX_INST_GROUP_BLOCKS), dim3(VGX_INST_GROUP_THREADS), 0, s, draws, ndraws, npaths, nc, (const VgxTotals*)totals,
			(const uint64_t*)(nc > 1 ? keyStart : start), cursor, order);
	}
}

// ---- periodic mode, instances of different scales: a permutation of the INSTANCES -------------------------------------
// Counting sort of the instances by the tolerance class of their first draw (inst_tol_bits quantised over the range of those
// draws, as in the grouped mode): ninst items, nc <= 1024 classes -- three small launches. Typical instancing gives every
// instance ONE scale for all of its paths, so lanes that agree on the first path agree on all of them; where they do not,
// a cubic falls back to the per-lane walk as always.
namespace {
__global__ __launch_bounds__(256) void k_inst_perm_range(const vgx_draw* draws, uint64_t ninst, uint32_t period, VgxTotals* totals)
{
	uint32_t lo = 0xFFFFFFFFu, hi = 0u;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ninst; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t b = inst_tol_bits(draws + i * period);
		lo = b < lo ? b : lo; hi = b > hi ? b : hi;
	}
	for (int o = 32; o > 0; o >>= 1) {
		const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, o), h2 = (uint32_t)__shfl_xor((int)hi, o);
		lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
	}
	if ((threadIdx.x & 63) == 0 && lo <= hi) { atomicMax(&totals->inst_tol_lo_inv, ~lo); atomicMax(&totals->inst_tol_hi, hi); }
}
__device__ __forceinline__ uint32_t inst_perm_class(const vgx_draw* d, const InstKeys& K)
{
	const uint32_t b = inst_tol_bits(d);
	uint32_t c = b > K.lo ? (b - K.lo) >> K.shift : 0u;
	return c < K.nc ? c : K.nc - 1u;
}
__global__ __launch_bounds__(256) void k_inst_perm_hist(const vgx_draw* draws, uint64_t ninst, uint32_t period, uint32_t nc, const VgxTotals* totals, uint32_t* hist)
{
	const InstKeys K = inst_keys(1u, nc, totals);
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ninst; i += (uint64_t)gridDim.x * blockDim.x) {
		atomicAdd(&hist[inst_perm_class(draws + i * period, K)], 1u);
	}
}
// one workgroup: exclusive scan of the class histogram in place (nc <= 1024)
__global__ __launch_bounds__(1024) void k_inst_perm_scan(uint32_t nc, uint32_t* hist)
{
	__shared__ uint32_t s[1024];
	const uint32_t t = threadIdx.x;
	const uint32_t v = t < nc ? hist[t] : 0u;
	s[t] = v;
	__syncthreads();
	for (uint32_t o = 1; o < 1024; o <<= 1) {
		const uint32_t x = t >= o ? s[t - o] : 0u;
		__syncthreads();
		s[t] += x;
		__syncthreads();
	}
	if (t < nc) { hist[t] = s[t] - v; }
}
__global__ __launch_bounds__(256) void k_inst_perm_scatter(const vgx_draw* draws, uint64_t ninst, uint32_t period, uint32_t nc, const VgxTotals* totals, uint32_t* cursor, uint32_t* perm)
{
	const InstKeys K = inst_keys(1u, nc, totals);
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ninst; i += (uint64_t)gridDim.x * blockDim.x) {
		perm[atomicAdd(&cursor[inst_perm_class(draws + i * period, K)], 1u)] = (uint32_t)i;
	}
}
}

// up to 2^18 instances: the whole sort in ONE workgroup (range -> LDS histogram -> scan -> scatter): four dependent launches of
// a handful of workgroups each cost 0.09 ms for 10 000 instances, this one 0.02
__global__ __launch_bounds__(1024) void k_inst_perm_single(const vgx_draw* draws, uint32_t ninst, uint32_t period, uint32_t nc, uint32_t* perm)
{
	__shared__ uint32_t s_hist[1024];
	__shared__ uint32_t s_lo, s_hi;
	const uint32_t t = threadIdx.x;
	if (t == 0) { s_lo = 0xFFFFFFFFu; s_hi = 0u; }
	s_hist[t] = 0;
	__syncthreads();
	uint32_t lo = 0xFFFFFFFFu, hi = 0u;
	for (uint32_t i = t; i < ninst; i += 1024) {
		const uint32_t b = inst_tol_bits(draws + (uint64_t)i * period);
		lo = b < lo ? b : lo; hi = b > hi ? b : hi;
	}
	for (int o = 32; o > 0; o >>= 1) {
		const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, o), h2 = (uint32_t)__shfl_xor((int)hi, o);
		lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
	}
	if ((t & 63) == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
	__syncthreads();
	InstKeys K;
	K.npaths = 1; K.nc = nc; K.lo = s_lo; K.shift = 0;
	{ const uint32_t span = s_hi > s_lo ? s_hi - s_lo : 0u; while ((span >> K.shift) >= nc) { ++K.shift; } }
	for (uint32_t i = t; i < ninst; i += 1024) { atomicAdd(&s_hist[inst_perm_class(draws + (uint64_t)i * period, K)], 1u); }
	__syncthreads();
	const uint32_t v = s_hist[t];
	__syncthreads();
	for (uint32_t o = 1; o < 1024; o <<= 1) {
		const uint32_t x = t >= o ? s_hist[t - o] : 0u;
		__syncthreads();
		s_hist[t] += x;
		__syncthreads();
	}
	const uint32_t excl = s_hist[t] - v;
	__syncthreads();
	s_hist[t] = excl; // running cursors
	__syncthreads();
	for (uint32_t i = t; i < ninst; i += 1024) { perm[atomicAdd(&s_hist[inst_perm_class(draws + (uint64_t)i * period, K)], 1u)] = i; }
}

void vgx_launch_inst_perm(const vgx_draw* draws, uint64_t ninst, uint32_t period, uint32_t nc, uint32_t* classHist, uint32_t* perm, VgxTotals* totals, bool multi, hipStream_t s)
{
	if (nc > 1024u) { nc = 1024u; }
	if (ninst <= (1ull << 18) && !multi) {
		hipLaunchKernelGGL(k_inst_perm_single, dim3(1), dim3(1024), 0, s, draws, (uint32_t)ninst, period, nc, perm);
		return;
	}
	(void)hipMemsetAsync(classHist, 0, ((size_t)nc + 1) * sizeof(uint32_t), s);
	const int blocks = (int)((ninst + 255) / 256 < 256 ? (ninst + 255) / 256 : 256);
	hipLaunchKernelGGL(k_inst_perm_range, dim3(blocks ? blocks : 1), dim3(256), 0, s, draws, ninst, period, totals);
	hipLaunchKernelGGL(k_inst_perm_hist, dim3(blocks ? blocks : 1), dim3(256), 0, s, draws, ninst, period, nc, (const VgxTotals*)totals, classHist);
	hipLaunchKernelGGL(k_inst_perm_scan, dim3(1), dim3(1024), 0, s, nc, classHist);
	hipLaunchKernelGGL(k_inst_perm_scatter, dim3(blocks ? blocks : 1), dim3(256), 0, s, draws, ninst, period, nc, (const VgxTotals*)totals, classHist, perm);
}

void vgx_launch_flatten_inst(const VgxFlattenArgs& a, int waves, hipStream_t s)
{
	hipLaunchKernelGGL(k_flatten_inst, dim3(waves), dim3(VGX_WAVE), 0, s, a);
}

void vgx_launch_inst_detect(const vgx_draw* draws, uint64_t ndraws, VgxTotals* totals, hipStream_t s)
{
	if (ndraws < 2) { return; }
	hipLaunchKernelGGL(k_inst_find, dim3(512), dim3(256), 0, s, draws, ndraws, totals);
	hipLaunchKernelGGL(k_inst_verify, dim3(512), dim3(256), 0, s, draws, ndraws, totals);
	hipLaunchKernelGGL(k_inst_tol_range, dim3(512), dim3(256), 0, s, draws, ndraws, totals);
}
