// vgx_pathset.hip -- the derived tables of a path set, built on the device (round 6).
//
// vgx_pathset_create used to walk every command on ONE host thread five times (grammar, sub-path ends, a 64-byte record per
// command, the thin records, the static polyline layout): 75-96 ms for BASELINE configs[1]'s 2 M commands, 330-400 ms for
// configs[3]'s 10 M -- more than the 16-core reference needs for the whole batch. Here the caller's four raw arrays are uploaded
// as they are and everything else is a handful of HBM-bound passes over them:
//
//   k_ps_front       one lane per path       path_cmd_begin monotone? first command of every non-empty path marked (pathAt),
//                                            longest path, empty paths
//                    one lane per argument   finite?
//   scan A (3 passes, vgx_mscan.h) over commands, monoid (max, max, +): per command from its own / its neighbours' opcodes --
//                    the reference's "is a sub-path open" state is LOCAL: open after command c = !(CLOSE or a closed shape),
//                    so STARTS_SUB / LAST_IN_SUB / NEXT_IS_CLOSE / LAST_IN_PATH and every grammar check need only c - 1, c, c + 1
//                    -- and by the scan: the sub-path's first command (cmd_sp_start), the path of the command, the number of
//                    sub-paths in front (-> path_sub_begin, sub_last_cmd, the sub-path ordinal)
//   k_ps_back        one lane per command    VgxCmdRec (64 B: arguments, start point = the previous command's last pair, the
//                                            sub-path's first point: gathers), VgxCmdThin, sub_last_cmd
//                    one lane per path       path_sub_begin, VGX_PF_THIN
//   thin sets only (every path MOVE_TO / LINE_TO / CLOSE; vgx_thin.h):
//   scan B (vgx_scan.h) over commands        vertices in front of the command (+1 / 0 / -1 for the vertex pathClose pops, path.cpp:716-725)
//   k_ps_thin_path, k_ps_thin_cmd            VgxThinPath, VgxThinSub, the vertex place and sub-path ordinal of every command
//
// Nothing here decides a status code: any failed check sets VgxPsTotals::err and the host validator (vgx_pathset_validate, the
// slow path of an invalid set) names it. No host round trip between the kernels; the host reads VgxPsTotals once at the end.
#include "vgx_pathset_dev.h"
#include "vgx_mscan.h"
#include "vgx_wave.h"
#include "../../include/vgx.h"

namespace {

__device__ __forceinline__ bool ps_is_shape(uint32_t t) { return t >= VGX_CMD_RECT && t <= VGX_CMD_ELLIPSE; }
__device__ __forceinline__ bool ps_closes(uint32_t t) { return t == VGX_CMD_CLOSE || ps_is_shape(t); } // no open sub-path behind it
__device__ __forceinline__ int ps_arg_count(uint32_t t) // kArgCount of vgx_pathset_host.h, a nibble per opcode
{
	return (int)((0x438546504622ull >> (4u * t)) & 0xFull);
}

struct VgxPsMOps : VgxPsM
{
	static __device__ __forceinline__ VgxPsMOps identity() { VgxPsMOps r; r.head1 = 0; r.path1 = 0; r.nlast = 0; r.f = 0; return r; }
	static __device__ __forceinline__ VgxPsMOps combine(VgxPsMOps a, VgxPsMOps b)
	{
		VgxPsMOps r;
		r.head1 = a.head1 > b.head1 ? a.head1 : b.head1; r.path1 = a.path1 > b.path1 ? a.path1 : b.path1; r.nlast = a.nlast + b.nlast; r.f = a.f | b.f;
		return r;
	}
	static __device__ __forceinline__ VgxPsMOps shfl_up(VgxPsMOps v, int d)
	{
		VgxPsMOps r;
		r.head1 = __shfl_up(v.head1, d); r.path1 = __shfl_up(v.path1, d); r.nlast = __shfl_up(v.nlast, d); r.f = v.f;
		return r;
	}
};

__device__ __forceinline__ void ps_mark(const VgxPsBuild& A, uint32_t p)
{
	uint32_t len = 0;
	bool empty = false, bad = false;
	if (p < A.npaths) {
		const uint32_t c0 = A.pcb[p], c1 = A.pcb[p + 1];
		if (c1 < c0 || c1 > A.ncmd) { bad = true; }
		else if (c1 > c0) { A.pathAt[c0] = p + 1u; len = c1 - c0; }
		else { empty = true; }
	}
	// one atomic per wave, not per path (a million two-command paths would queue a million atomics on one address)
	for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(len, d); len = o > len ? o : len; }
	const uint64_t anyEmpty = wave_ballot(empty), anyBad = wave_ballot(bad);
	if ((threadIdx.x & 63) == 0) {
		if (len) { atomicMax(&A.tot->maxCmds, len); }
		if (anyEmpty) { A.tot->hasEmpty = 1u; }
		if (anyBad) { A.tot->err = 1u; }
	}
}

__device__ __forceinline__ void ps_args(const VgxPsBuild& A, uint32_t first, uint32_t stride)
{
	bool bad = false;
	for (uint32_t i = first; i < A.nargs; i += stride) {
		const uint32_t u = __float_as_uint(A.args[i]);
		bad |= (u & 0x7F800000u) == 0x7F800000u; // !isfinite
	}
	if (wave_ballot(bad) && (threadIdx.x & 63) == 0) { A.tot->err = 1u; }
}

struct OpPsA
{
	VgxPsBuild A;
	__device__ uint32_t size() const { return A.ncmd; }
	__device__ VgxPsMOps load(uint32_t c) const
	{
		VgxPsMOps r = VgxPsMOps::identity();
		uint32_t t = A.type[c];
		bool bad = false;
		if (t >= VGX_CMD_COUNT_) { bad = true; t = VGX_CMD_MOVE_TO; }
		const uint32_t pa = A.pathAt[c];
		const bool first = pa != 0u;
		const bool last = (c + 1u == A.ncmd) || A.pathAt[c + 1u] != 0u;
		const uint32_t ao = A.argOff[c], ao1 = A.argOff[c + 1u];
		if (ao1 < ao || ao1 > A.nargs) { bad = true; }
		else {
			const uint32_t na = ao1 - ao;
			if (t == VGX_CMD_POLYLINE) { bad |= na < 2u || (na & 1u); }
			else if ((int)na != ps_arg_count(t)) { bad = true; }
			else if (t == VGX_CMD_ARC) { // the angle-wrapping loops of pathArc (path.cpp:637-652) stay short below 1e5
				bad |= fabsf(A.args[ao + 3u]) > 1.0e5f || fabsf(A.args[ao + 4u]) > 1.0e5f;
			}
		}
		uint32_t tp = VGX_CMD_CLOSE; // "nothing open in front" at the start of a path
		if (!first) { tp = A.type[c - 1u]; if (tp >= VGX_CMD_COUNT_) { tp = VGX_CMD_MOVE_TO; } }
		const bool openBefore = !first && !ps_closes(tp);
		bool starts = false;
		if (t == VGX_CMD_MOVE_TO || ps_is_shape(t)) { starts = true; }
		else if (t == VGX_CMD_ARC) { starts = !openBefore; bad |= !openBefore && !first; } // pathArc: moveTo when nothing is open (path.cpp:663-667)
		else if (!openBefore) { bad = true; } // LINE_TO / CUBIC_TO / ... need an open sub-path (path.cpp:82,88)
		uint32_t fl = starts ? VGX_CF_STARTS_SUB : 0u;
		if (last) { fl |= VGX_CF_LAST_IN_PATH | VGX_CF_LAST_IN_SUB; }
		else {
			uint32_t tn = A.type[c + 1u];
			if (tn >= VGX_CMD_COUNT_) { tn = VGX_CMD_MOVE_TO; }
			const bool nextStarts = tn == VGX_CMD_MOVE_TO || ps_is_shape(tn) || (tn == VGX_CMD_ARC && ps_closes(t));
			if (nextStarts) { fl |= VGX_CF_LAST_IN_SUB; }
			if (tn == VGX_CMD_CLOSE) { fl |= VGX_CF_NEXT_IS_CLOSE; }
		}
		if (bad) { A.tot->err = 1u; }
		const bool serial = t == VGX_CMD_ARC || t == VGX_CMD_ARC_TO || ps_is_shape(t);
		const bool notThin = !(t == VGX_CMD_MOVE_TO || t == VGX_CMD_LINE_TO || t == VGX_CMD_CLOSE);
		r.head1 = starts ? c + 1u : 0u;
		r.path1 = pa;
		r.nlast = (fl & VGX_CF_LAST_IN_SUB) ? 1u : 0u;
		r.f = fl | (t << 8) | (serial ? 0x10000u : 0u) | (notThin ? 0x20000u : 0u);
		return r;
	}
	__device__ void store(uint32_t c, VgxPsMOps incl, VgxPsMOps own) const
	{
		A.spStart[c] = incl.head1 ? incl.head1 - 1u : c;
		const uint32_t p = incl.path1 ? incl.path1 - 1u : 0u;
		A.pathOf[c] = p;
		A.lastSubEx[c] = incl.nlast - own.nlast;
		A.flags[c] = (uint8_t)(own.f & 0xFFu);
		if (own.f & 0x30000u) { // rare for polylines, once per command for curves: a byte-wide OR into the path's flag
			const uint32_t bits = ((own.f & 0x10000u) ? VGX_PF_SERIAL : 0u) | ((own.f & 0x20000u) ? 0x80u : 0u);
			atomicOr((uint32_t*)(A.pathFlags + (p & ~3u)), bits << (8u * (p & 3u)));
			if (own.f & 0x10000u) { A.tot->hasSerial = 1u; }
			A.tot->notThin = 1u;
		}
	}
	__device__ void finish(VgxPsMOps total) const { A.lastSubEx[A.ncmd] = total.nlast; A.tot->nsubs = total.nlast; }
};

__device__ __forceinline__ void ps_rec(const VgxPsBuild& A, uint32_t c)
{
	if (c >= A.ncmd || A.tot->err) { return; } // (an invalid set: its offsets may point anywhere; the build ends without tables)
	const uint32_t t = A.type[c], fl = A.flags[c];
	const uint32_t ao = A.argOff[c];
	const uint32_t na = A.argOff[c + 1u] - ao;
	float a[8];
#pragma unroll
	for (uint32_t i = 0; i < 8u; ++i) { a[i] = (i < na && t != VGX_CMD_POLYLINE) ? A.args[ao + i] : 0.0f; }
	const float sx = ao >= 2u ? A.args[ao - 2u] : 0.0f, sy = ao >= 2u ? A.args[ao - 1u] : 0.0f;
	float hx = 0.0f, hy = 0.0f; // first point of the command's sub-path when a MOVE_TO opened it
	{
		const uint32_t hc = A.spStart[c];
		if (A.type[hc] == VGX_CMD_MOVE_TO) {
			const uint32_t ho = A.argOff[hc];
			hx = A.args[ho]; hy = A.args[ho + 1u];
			// pathClose's last-vs-first test (path.cpp:716-722) is evaluated by the CLOSE lane and by the lane in front of it
			if (t <= VGX_CMD_CLOSE || t == VGX_CMD_POLYLINE) { a[6] = hx; a[7] = hy; }
		}
	}
	float4* r = (float4*)(A.rec + c);
	r[0] = make_float4(__uint_as_float(t), __uint_as_float(fl), __uint_as_float(na), __uint_as_float(ao));
	r[1] = make_float4(sx, sy, a[0], a[1]);
	r[2] = make_float4(a[2], a[3], a[4], a[5]);
	r[3] = make_float4(a[6], a[7], 0.0f, 0.0f);
	float tx = 0.0f, ty = 0.0f;
	if (t == VGX_CMD_MOVE_TO || t == VGX_CMD_LINE_TO) { tx = A.args[ao]; ty = A.args[ao + 1u]; }
	else if (t == VGX_CMD_CLOSE) { tx = hx; ty = hy; }
	*(float4*)(A.thin + c) = make_float4(__uint_as_float(t | (fl << 8)), tx, ty, __uint_as_float(0u));
	if (fl & VGX_CF_LAST_IN_SUB) { A.subLast[A.lastSubEx[c]] = c - A.pcb[A.pathOf[c]]; }
}

__device__ __forceinline__ void ps_pathfix(const VgxPsBuild& A, uint32_t p)
{
	if (p > A.npaths || A.tot->err) { return; }
	if (p == A.npaths) { A.subBegin[p] = A.ncmd ? A.lastSubEx[A.ncmd] : 0u; return; }
	const uint32_t c0 = A.pcb[p], c1 = A.pcb[p + 1];
	A.subBegin[p] = A.ncmd ? A.lastSubEx[c0] : 0u; // (an empty path: the count in front of the next command, or the total)
	uint32_t f = A.pathFlags[p];
	const bool thin = c1 > c0 && !(f & VGX_PF_SERIAL) && !(f & 0x80u);
	A.pathFlags[p] = (uint8_t)((f & ~0x80u & ~VGX_PF_THIN) | (thin ? VGX_PF_THIN : 0u));
}

__device__ __forceinline__ bool ps_all_thin(const VgxPsBuild& A)
{
	return A.npaths != 0u && A.ncmd != 0u && !A.tot->notThin && !A.tot->hasSerial && !A.tot->hasEmpty && !A.tot->err;
}

// vertices command c adds to its path's polyline: vgx_thin_build's `cnt` (MOVE_TO / LINE_TO: one; CLOSE: minus one when it pops)
__device__ __forceinline__ int ps_thin_cnt(const VgxPsBuild& A, uint32_t c, bool* closedHere)
{
	const VgxCmdThin t = A.thin[c];
	const uint32_t type = t.meta & 0xFFu;
	*closedHere = false;
	if (type != VGX_CMD_CLOSE) { return 1; }
	const uint32_t sp = c - A.spStart[c]; // vertices of the open sub-path: its MOVE_TO and the LINE_TOs behind it
	if (sp > 2u) { // pathClose, path.cpp:707-726
		*closedHere = true;
		const VgxCmdThin q = A.thin[(int64_t)c - 1];
		if (v2near(v2(q.x, q.y), v2(t.x, t.y))) { return -1; }
	}
	return 0;
}

struct OpPsB // exclusive sum of ps_thin_cnt over the commands (vgx_scan.h; sums are taken modulo 2^64, every prefix is >= 0)
{
	VgxPsBuild A;
	__device__ uint64_t size() const { return ps_all_thin(A) ? (uint64_t)A.ncmd : 0ull; }
	__device__ Sum3 load(uint64_t i) const { Sum3 r = sum3_zero(); bool cl; r.a = (uint64_t)(int64_t)ps_thin_cnt(A, (uint32_t)i, &cl); return r; }
	__device__ void store(uint64_t i, Sum3 e) const { A.nvEx[i] = (uint32_t)e.a; }
	__device__ void finish(Sum3 t) const { A.nvEx[A.ncmd] = (uint32_t)t.a; }
};

__device__ __forceinline__ void ps_thin_path(const VgxPsBuild& A, uint32_t p)
{
	if (p >= A.npaths) { return; }
	const uint32_t c0 = A.pcb[p], c1 = A.pcb[p + 1];
	VgxThinPath q;
	q.pc0 = c0; q.nverts = A.nvEx[c1] - A.nvEx[c0]; q.nsubs = A.subBegin[p + 1] - A.subBegin[p]; q.nge3 = 0; q.nge2 = 0; q.flags = 0; q.sub0 = A.subBegin[p]; q.pad = 0;
	if (q.nsubs > 65536u) { A.tot->thinIneligible = 1u; }
	A.tp[p] = q;
}
__global__ __launch_bounds__(256) void k_ps_thin_path(VgxPsBuild A) { if (ps_all_thin(A)) { ps_thin_path(A, blockIdx.x * 256u + threadIdx.x); } }

__device__ __forceinline__ void ps_thin_cmd(const VgxPsBuild& A, uint32_t c)
{
	if (c >= A.ncmd) { return; }
	const uint32_t p = A.pathOf[c];
	const uint32_t c0 = A.pcb[p];
	const VgxCmdThin t = A.thin[c];
	const uint32_t type = t.meta & 0xFFu, fl = (t.meta >> 8) & 0xFFu;
	bool closedHere;
	const int cnt = ps_thin_cnt(A, c, &closedHere);
	const uint32_t nv = A.nvEx[c] - A.nvEx[c0];
	const uint32_t j = A.lastSubEx[c] - A.lastSubEx[c0]; // sub-paths of the path in front of this command's = its ordinal
	uint32_t place = VGX_THIN_NONE;
	if (cnt == 1) {
		place = nv;
		if (!(fl & VGX_CF_LAST_IN_PATH) && (fl & VGX_CF_NEXT_IS_CLOSE)) { // popped by the pathClose behind it: never stored
			bool cl2;
			if (ps_thin_cnt(A, c + 1u, &cl2) == -1) { place = VGX_THIN_NONE; }
		}
	}
	if (type == VGX_CMD_LINE_TO) {
		const VgxCmdThin q = A.thin[(int64_t)c - 1];
		if (v2near(v2(q.x, q.y), v2(t.x, t.y))) { atomicOr(&A.tp[p].flags, VGX_THIN_DEGENERATE); } // pathLineTo drops it (path.cpp:769-775): the exact builder takes the draw
	}
	VgxCmdThin o = t;
	o.meta = (t.meta & 0xFFFFu) | ((j & 0xFFFFu) << 16);
	o.pad = place;
	*(float4*)(A.thin + c) = make_float4(__uint_as_float(o.meta), o.x, o.y, __uint_as_float(o.pad));
	if (fl & VGX_CF_LAST_IN_SUB) {
		const uint32_t sp = c - A.spStart[c]; // vertices of the sub-path in front of this command
		const uint32_t spTotal = (uint32_t)((int)sp + cnt);
		VgxThinSub s;
		s.first = nv - sp; s.info = spTotal | (closedHere ? 0x80000000u : 0u);
		A.ts[A.lastSubEx[c]] = s;
		if (spTotal >= 3u) { atomicAdd(&A.tp[p].nge3, 1u); }
		if (spTotal >= 2u) { atomicAdd(&A.tp[p].nge2, 1u); }
	}
}
__global__ __launch_bounds__(256) void k_ps_thin_cmd(VgxPsBuild A) { if (ps_all_thin(A)) { ps_thin_cmd(A, blockIdx.x * 256u + threadIdx.x); } }

// Fused launches (a frame-sized set is bound by the NUMBER of dependent launches, each ~6 us on a GPU that idles between frames: a
// single-workgroup kernel running every pass behind block barriers was built and measured -- 134 us per Tiger frame against 111 us
// for the separate launches: one compute unit at idle clocks is slower than a dozen launches that each use the chip):
//   k_ps_front  paths -> ps_mark; arguments -> ps_args; and the bytes the later passes OR into / the padding records the flatten
//               kernels read past the ends (no memset of the blob)
//   k_ps_back   commands -> ps_rec; paths -> ps_pathfix
__global__ __launch_bounds__(256) void k_ps_front(VgxPsBuild A, uint32_t markBlocks)
{
	if (blockIdx.x < markBlocks) {
		const uint32_t p = blockIdx.x * 256u + threadIdx.x;
		ps_mark(A, p);
		if (p < (A.npaths + 4u) / 4u) { ((uint32_t*)A.pathFlags)[p] = 0u; } // (the blob keeps npaths + 4 bytes, 256-byte aligned)
		if (p == 0) {
			const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			*(float4*)(A.thin - 1) = z; *(float4*)(A.thin + A.ncmd) = z; *(float4*)(A.thin + A.ncmd + 1u) = z;
			float4* r = (float4*)(A.rec + A.ncmd);
			r[0] = z; r[1] = z; r[2] = z; r[3] = z;
			A.spStart[A.ncmd] = 0u; A.flags[A.ncmd] = 0;
			((float*)A.args)[-2] = 0.0f; ((float*)A.args)[-1] = 0.0f;
			((uint8_t*)A.type)[A.ncmd] = 0;
		}
	} else {
		ps_args(A, (blockIdx.x - markBlocks) * 256u + threadIdx.x, (gridDim.x - markBlocks) * 256u);
	}
}

__global__ __launch_bounds__(256) void k_ps_back(VgxPsBuild A, uint32_t recBlocks)
{
	if (blockIdx.x < recBlocks) { ps_rec(A, blockIdx.x * 256u + threadIdx.x); }
	else { ps_pathfix(A, (blockIdx.x - recBlocks) * 256u + threadIdx.x); }
}

} // namespace

// maybeThin: false when the host knows that some command is not MOVE_TO / LINE_TO / CLOSE (it looks at the opcodes of frame-sized
// sets: a few KB); the thin passes then are not launched at all (they would find that out themselves and exit).
void vgx_launch_pathset_build(const VgxPsBuild& a, bool maybeThin, hipStream_t s)
{
	const uint32_t markBlocks = (a.npaths + 1u + 255u) / 256u; // >= 1: block 0 also writes the padding records
	uint32_t argBlocks = (a.nargs + 255u) / 256u;
	if (argBlocks > 4096u) { argBlocks = 4096u; }
	hipLaunchKernelGGL(k_ps_front, dim3(markBlocks + argBlocks), dim3(256), 0, s, a, markBlocks);
	const uint32_t recBlocks = (a.ncmd + 255u) / 256u;
	if (a.ncmd) {
		OpPsA opA; opA.A = a;
		vgx_monoid_scan<VgxPsMOps, OpPsA>(opA, (VgxPsMOps*)a.partialA, s);
	}
	hipLaunchKernelGGL(k_ps_back, dim3(recBlocks + (a.npaths + 1u + 255u) / 256u), dim3(256), 0, s, a, recBlocks);
	if (a.ncmd && a.npaths && maybeThin) {
		OpPsB opB; opB.A = a;
		vgx_device_scan(opB, a.partialB, s, a.ncmd);
		hipLaunchKernelGGL(k_ps_thin_path, dim3((a.npaths + 255u) / 256u), dim3(256), 0, s, a);
		hipLaunchKernelGGL(k_ps_thin_cmd, dim3((a.ncmd + 255u) / 256u), dim3(256), 0, s, a);
	}
}
