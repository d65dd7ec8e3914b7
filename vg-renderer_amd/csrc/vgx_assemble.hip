// vgx_assemble.hip -- vertex-buffer / draw-command assembly on gfx950.
//
// Replaces the bookkeeping the reference does per mesh in createDrawCommand_VertexColor (src/vg.cpp:5207-5244) through
// allocVertices (:5321-5342), allocIndices (:5344-5357) and allocDrawCommand (:5359-5407) for a frame whose meshes
// all share one draw state (type Textured / font image 0 / one scissor / no clip change -- what a run of ctxFillPath /
// ctxStrokePath colour calls produces):
//   - vertices go to the current vertex buffer until `count + numVertices > maxVBVertices`, then a new vertex buffer
//     starts (allocVertices) and with it a new draw command (m_ForceNewDrawCommand); otherwise the mesh merges into the
//     previous command (same type and handle, :5376-5379);
//   - indices go to the frame's single index buffer, rebased by the vertices already in the draw command:
//     dst = src + (uint16_t)cmd->m_NumVertices (vgutil::batchTransformDrawIndices, vg_util.cpp:447-520).
// The vertex streams of the tessellator are already in vertex-buffer order, so assembly = (1) the greedy partition of
// the mesh sequence into vertex buffers, (2) one uint32 base per mesh that k_fill / k_stroke add to every index they
// write (no extra pass over the 2-byte index stream), (3) the draw-command table.
//
// (1) is a sequential recurrence in the reference: start(t+1) = first mesh that does not fit after start(t). Here:
//   k_asm_next   next[i] = first mesh j > i with V[j] + nv[j] - V[i] > maxVB (binary search in the vertex prefix V),
//                for every mesh i in parallel -- "if a vertex buffer started at i, where would the following one start";
//   k_asm_round  pointer doubling: with J = next^(n) as a table and the first n buffer starts known,
//                start[n + t] = J[start[t]] for t < n, then J <- J o J. log2(#buffers) rounds;
//   k_asm_assign per mesh: its vertex buffer by binary search in start[], base = V[i] - V[start], and one thread per
//                buffer writes the draw command.
// With VGX_ASM_SPLIT_STATE a draw command also ends where the draws' state_key changes between consecutive meshes (type /
// handle mismatch in allocDrawCommand, vg.cpp:5376-5379; a clip command list or a forced new command is a key change the
// host folded into the word, include/vgx.h): command starts = vertex-buffer starts UNION key changes. After the chain
// of vertex buffers is known, k_asm_vb marks every mesh that starts a vertex buffer, a device scan over the start flags
// numbers the commands, and k_asm_cmd_finish fills the per-mesh index base (vertices in front of the mesh inside its
// COMMAND, the value the reference rebases by) and the command sizes.
// The white-pixel UV stream of createDrawCommand_VertexColor (vg.cpp:5218-5225) is one constant per vertex: k_asm_uv.
#include "vgx_internal.h"
#include "vgx_scan.h"

namespace {

// first j in (i, M] with V(j) + nv(j) - V(i) > maxVB, where mesh M is a sentinel that never fits.
// The answer moves with i (next[i] - i ~ maxVB / average mesh size), so the search gallops outwards from that estimate
// and finishes with a binary search inside the bracket: ~2 log2(error) probes that neighbouring threads share in cache,
// instead of log2(M) scattered ones.
__device__ __forceinline__ void asm_next(const VgxAsmArgs& A, uint64_t tid, uint64_t stride)
{
	const uint64_t M = A.totals->sizes.num_meshes;
	const uint64_t totalV = A.totals->sizes.num_vertices;
	const uint64_t est = M && totalV ? (uint64_t)A.max_vb * M / totalV : 1; // meshes per vertex buffer, on average
	for (uint64_t i = tid; i <= M; i += stride) {
		if (i == M) { A.jump0[M] = (uint32_t)M; continue; }
		const uint64_t limit = A.mtab[i].first_vertex + (uint64_t)A.max_vb;
		// meshes i..j-1 fit iff end(j-1) = V[j-1] + nv[j-1] <= limit; ends are non-decreasing.
		// fits(j) := end(j) <= limit for j in [i+1, M); the answer is the first j in [i+1, M] that does not fit (M: none).
		uint64_t lo = i + 1, hi = M; // invariant: every j < lo fits, hi does not fit (or hi == M)
		uint64_t g = i + (est > 0 ? est : 1);
		if (g < lo) { g = lo; }
		if (g < hi) {
			const vgx_mesh m = A.mtab[g];
			if (m.first_vertex + m.num_vertices <= limit) { // estimate fits: gallop upwards
				lo = g + 1;
				uint64_t step = 1;
				while (lo < hi) {
					const uint64_t p = lo + step - 1 < hi ? lo + step - 1 : hi - 1;
					const vgx_mesh q = A.mtab[p];
					if (q.first_vertex + q.num_vertices <= limit) { lo = p + 1; step <<= 1; } else { hi = p; break; }
				}
			} else { // gallop downwards
				hi = g;
				uint64_t step = 1;
				while (lo < hi) {
					const uint64_t p = hi - lo > step ? hi - step : lo;
					const vgx_mesh q = A.mtab[p];
					if (q.first_vertex + q.num_vertices <= limit) { lo = p + 1; break; } else { hi = p; step <<= 1; }
				}
			}
		}
		while (lo < hi) {
			const uint64_t mid = (lo + hi) >> 1;
			const vgx_mesh m = A.mtab[mid];
			if (m.first_vertex + m.num_vertices <= limit) { lo = mid + 1; } else { hi = mid; }
		}
		A.jump0[i] = (uint32_t)lo;
	}
}
__global__ __launch_bounds__(256) void k_asm_next(VgxAsmArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	asm_next(A, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// one doubling round: n = number of buffer starts known so far (a power of two), J = next^n
__device__ __forceinline__ void asm_round(const VgxAsmArgs& A, const uint32_t* J, uint32_t* Jout, uint32_t n, uint64_t tid, uint64_t stride)
{
	const uint64_t M = A.totals->sizes.num_meshes;
	if (M == 0 || A.start[n - 1] >= M) { return; } // the chain already ran off the end: nothing left to extend (uniform: start[n - 1] was written a round ago)
	for (uint64_t i = tid; i <= M; i += stride) {
		if (i < n) {
			const uint32_t s = A.start[i];
			A.start[n + i] = s >= M ? (uint32_t)M : J[s];
		}
		Jout[i] = J[J[i]];
	}
}
__global__ __launch_bounds__(256) void k_asm_round(VgxAsmArgs A, const uint32_t* J, uint32_t* Jout, uint32_t n)
{
	if (A.totals->status != VGX_OK) { return; }
	asm_round(A, J, Jout, n, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

__global__ __launch_bounds__(256) void k_asm_init(VgxAsmArgs A)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < A.cap_start) { A.start[i] = i == 0 ? 0u : 0xFFFFFFFFu; }
}

__device__ __forceinline__ void asm_assign(const VgxAsmArgs& A, const uint64_t tid, const uint64_t stride)
{
	const uint64_t M = A.totals->sizes.num_meshes;
	// number of vertex buffers T = entries of start[] below M (start[] is increasing until it saturates at M / unset)
	uint64_t lo = 0, hi = A.cap_start;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if ((uint64_t)A.start[mid] < M) { lo = mid + 1; } else { hi = mid; }
	}
	const uint64_t T = lo;
	if (tid == 0) {
		A.totals->sizes.num_drawcmds = T;
		if (T > A.cap_drawcmds) { atomicCAS(&A.totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_NOSPACE); }
		if (A.dev_num_drawcmds) { *A.dev_num_drawcmds = T; }
	}
	if (T > A.cap_drawcmds) { return; }
	for (uint64_t i = tid; i < M; i += stride) {
		uint64_t a = 0, b = T; // last t with start[t] <= i
		while (b - a > 1) {
			const uint64_t mid = (a + b) >> 1;
			if ((uint64_t)A.start[mid] <= i) { a = mid; } else { b = mid; }
		}
		const uint64_t s = A.start[a];
		const vgx_mesh mi = A.mtab[i];
		A.mesh_base[i] = (uint32_t)(mi.first_vertex - A.mtab[s].first_vertex);
		if (mi.num_vertices > A.max_vb) { atomicCAS(&A.totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_MESH_TOO_LARGE); }
		if (i == s) { // first mesh of vertex buffer a: its draw command
			const uint64_t e = (a + 1 < T) ? (uint64_t)A.start[a + 1] : M;
			const uint64_t endV = (e < M) ? A.mtab[e].first_vertex : A.totals->sizes.num_vertices;
			const uint64_t endI = (e < M) ? A.mtab[e].first_index : A.totals->sizes.num_indices;
			vgx_drawcmd c;
			c.first_vertex = mi.first_vertex;
			c.first_index = mi.first_index;
			c.first_mesh = s;
			c.num_vertices = (uint32_t)(endV - mi.first_vertex);
			c.num_indices = (uint32_t)(endI - mi.first_index);
			c.num_meshes = (uint32_t)(e - s);
			c.vertex_buffer = (uint32_t)a;
			c.first_vertex_in_vb = 0;
			c.state_key = 0;
			A.drawcmds[a] = c;
		}
	}
}
__global__ __launch_bounds__(256) void k_asm_assign(VgxAsmArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	asm_assign(A, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// ---- VGX_ASM_SPLIT_STATE -------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t asm_num_vbs(const VgxAsmArgs& A, uint64_t M)
{
	uint64_t lo = 0, hi = A.cap_start; // entries of start[] below M (increasing until it saturates at M / unset)
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if ((uint64_t)A.start[mid] < M) { lo = mid + 1; } else { hi = mid; }
	}
	return lo;
}

// per mesh: its vertex buffer (bit 31: the mesh starts it). Written over jump1 (the doubling is done).
__device__ __forceinline__ void asm_vb(const VgxAsmArgs& A, uint32_t* meshVb, uint64_t tid, uint64_t stride)
{
	const uint64_t M = A.totals->sizes.num_meshes;
	const uint64_t T = asm_num_vbs(A, M);
	for (uint64_t i = tid; i < M; i += stride) {
		uint64_t a = 0, b = T; // last t with start[t] <= i
		while (b - a > 1) {
			const uint64_t mid = (a + b) >> 1;
			if ((uint64_t)A.start[mid] <= i) { a = mid; } else { b = mid; }
		}
		meshVb[i] = (uint32_t)a | ((uint64_t)A.start[a] == i ? 0x80000000u : 0u);
		if (A.mtab[i].num_vertices > A.max_vb) { atomicCAS(&A.totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_MESH_TOO_LARGE); }
	}
}
__global__ __launch_bounds__(256) void k_asm_vb(VgxAsmArgs A, uint32_t* meshVb)
{
	if (A.totals->status != VGX_OK) { return; }
	asm_vb(A, meshVb, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

struct OpAsmCmd // scan over the meshes of "this mesh starts a draw command"
{
	VgxAsmArgs A;
	const uint32_t* meshVb;
	int fixedSize; uint64_t fixedCount; // single-workgroup caller: the count every thread must agree on (read once, uniformly)
	__device__ uint64_t size() const { return fixedSize ? fixedCount : (A.totals->status == VGX_OK ? A.totals->sizes.num_meshes : 0); }
	__device__ uint32_t key(uint64_t i) const { return A.draws[A.mdesc[i].draw].state_key; }
	__device__ bool starts(uint64_t i) const { return (meshVb[i] >> 31) != 0 || i == 0 || key(i) != key(i - 1); }
	__device__ Sum3 load(uint64_t i) const { Sum3 r = sum3_zero(); r.a = starts(i) ? 1 : 0; return r; }
	__device__ void store(uint64_t i, Sum3 e) const
	{
		const bool st = starts(i);
		const uint64_t c = e.a + (st ? 1 : 0) - 1; // command of mesh i
		A.mesh_cmd[i] = (uint32_t)c;
		if (st && c < A.cap_drawcmds) {
			const uint32_t vb = meshVb[i] & 0x7FFFFFFFu;
			const vgx_mesh m = A.mtab[i];
			vgx_drawcmd d;
			d.first_vertex = m.first_vertex; d.first_index = m.first_index; d.first_mesh = i;
			d.num_vertices = 0; d.num_indices = 0; d.num_meshes = 0; // k_asm_cmd_finish
			d.vertex_buffer = vb;
			d.first_vertex_in_vb = (uint32_t)(m.first_vertex - A.mtab[A.start[vb]].first_vertex);
			d.state_key = key(i);
			A.drawcmds[c] = d;
		}
	}
	__device__ void finish(Sum3 t) const
	{
		A.totals->sizes.num_drawcmds = t.a;
		if (t.a > A.cap_drawcmds) { atomicCAS(&A.totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_NOSPACE); }
		if (A.dev_num_drawcmds) { *A.dev_num_drawcmds = t.a; }
	}
};

__device__ __forceinline__ void asm_cmd_finish(const VgxAsmArgs& A, uint64_t tid, uint64_t stride)
{
	const uint64_t M = A.totals->sizes.num_meshes;
	const uint64_t T = A.totals->sizes.num_drawcmds;
	if (T > A.cap_drawcmds) { return; } // (the table is full: VGX_E_NOSPACE is set, nothing past its end is touched)
	for (uint64_t i = tid; i < M; i += stride) {
		A.mesh_base[i] = (uint32_t)(A.mtab[i].first_vertex - A.drawcmds[A.mesh_cmd[i]].first_vertex);
	}
	for (uint64_t c = tid; c < T; c += stride) {
		vgx_drawcmd d = A.drawcmds[c];
		const bool last = c + 1 == T;
		const uint64_t endV = last ? A.totals->sizes.num_vertices : A.drawcmds[c + 1].first_vertex;
		const uint64_t endI = last ? A.totals->sizes.num_indices : A.drawcmds[c + 1].first_index;
		const uint64_t endM = last ? M : A.drawcmds[c + 1].first_mesh;
		A.drawcmds[c].num_vertices = (uint32_t)(endV - d.first_vertex);
		A.drawcmds[c].num_indices = (uint32_t)(endI - d.first_index);
		A.drawcmds[c].num_meshes = (uint32_t)(endM - d.first_mesh);
	}
}
__global__ __launch_bounds__(256) void k_asm_cmd_finish(VgxAsmArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	asm_cmd_finish(A, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// vgutil::memset32 / memset64 of the white-pixel UV over the frame's vertices (vg.cpp:5218-5225)
__device__ __forceinline__ void asm_uv(const VgxAsmArgs& A, const uint64_t tid, const uint64_t stride)
{
	const uint64_t n = A.totals->sizes.num_vertices;
	const uint64_t words = A.uv_bytes == 8 ? 2 * n : n; // 32-bit words
	const uint64_t quads = words / 4;
	uint32_t* p = (uint32_t*)A.uv;
	const uint32_t v0 = A.uv_value[0], v1 = A.uv_bytes == 8 ? A.uv_value[1] : A.uv_value[0];
	const uint4 q = make_uint4(v0, v1, v0, v1);
	if (((uintptr_t)p & 15u) == 0) {
		for (uint64_t i = tid; i < quads; i += stride) { ((uint4*)p)[i] = q; }
		for (uint64_t i = quads * 4 + tid; i < words; i += stride) { p[i] = (i & 1) ? v1 : v0; }
	} else {
		for (uint64_t i = tid; i < words; i += stride) { p[i] = (i & 1) ? v1 : v0; }
	}
}
__global__ __launch_bounds__(256) void k_asm_uv(VgxAsmArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	asm_uv(A, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// Frame-sized batches (a few hundred meshes, a handful of vertex buffers): every pass above by ONE workgroup, block barriers in
// place of the eight to ten dependent launches (each ~7 us on a GPU that idles between frames; the passes themselves are a few
// loads per mesh). The status is read ONCE, uniformly: a verdict reached inside (a mesh beyond any vertex buffer, the command table
// full) lets the remaining passes run to their ends -- they stay inside their tables -- and is published by the caller.
#define VGX_ASM_SMALL_THREADS 1024
__global__ __launch_bounds__(VGX_ASM_SMALL_THREADS) void k_asm_small(VgxAsmArgs A)
{
	__shared__ Sum3 s_wave[VGX_ASM_SMALL_THREADS / 64];
	__shared__ uint32_t s_status;
	__shared__ unsigned long long s_meshes;
	const uint64_t tid = threadIdx.x, T = VGX_ASM_SMALL_THREADS;
	if (tid == 0) {
		s_status = __hip_atomic_load(&A.totals->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		s_meshes = A.totals->sizes.num_meshes;
	}
	for (uint64_t i = tid; i < A.cap_start; i += T) { A.start[i] = i == 0 ? 0u : 0xFFFFFFFFu; }
	__syncthreads();
	if (s_status != VGX_OK) { return; }
	asm_next(A, tid, T);
	__syncthreads();
	const uint32_t* J = A.jump0;
	uint32_t* Jn = A.jump1;
	for (uint32_t n = 1; n < A.cap_start; n <<= 1) {
		asm_round(A, J, Jn, n, tid, T);
		__syncthreads();
		const uint32_t* t = J; J = Jn; Jn = (uint32_t*)t;
	}
	// (the host's loop swaps the tables the same number of times: jump1 holds dead doubling data at this point either way)
	if ((A.flags & VGX_ASM_SPLIT_STATE) && A.draws) {
		asm_vb(A, A.jump1, tid, T);
		__syncthreads();
		OpAsmCmd op;
		op.A = A; op.meshVb = A.jump1; op.fixedSize = 1; op.fixedCount = s_meshes;
		block_scan_all<OpAsmCmd, VGX_ASM_SMALL_THREADS>(op, s_wave);
		asm_cmd_finish(A, tid, T);
	} else {
		asm_assign(A, tid, T);
	}
	if (A.uv && A.uv_bytes) { asm_uv(A, tid, T); }
}

} // namespace

void vgx_launch_assemble(const VgxAsmArgs& a, hipStream_t s)
{
	if (a.scan_bound <= 8192 && a.cap_start <= 1024) { // a frame: one launch
		hipLaunchKernelGGL(k_asm_small, dim3(1), dim3(VGX_ASM_SMALL_THREADS), 0, s, a);
		return;
	}
	hipLaunchKernelGGL(k_asm_init, dim3((unsigned)((a.cap_start + 255) / 256)), dim3(256), 0, s, a);
	hipLaunchKernelGGL(k_asm_next, dim3(2048), dim3(256), 0, s, a);
	const uint32_t* J = a.jump0;
	uint32_t* Jn = a.jump1;
	for (uint32_t n = 1; n < a.cap_start; n <<= 1) { // doubles the known prefix of start[] every round
		hipLaunchKernelGGL(k_asm_round, dim3(2048), dim3(256), 0, s, a, J, Jn, n);
		const uint32_t* t = J; J = Jn; Jn = (uint32_t*)t;
	}
	if ((a.flags & VGX_ASM_SPLIT_STATE) && a.draws) {
		// the jump tables are dead now: jump1 holds every mesh's vertex buffer, jump0 (= mesh_cmd) its draw command
		hipLaunchKernelGGL(k_asm_vb, dim3(2048), dim3(256), 0, s, a, a.jump1);
		OpAsmCmd op;
		op.A = a; op.meshVb = a.jump1; op.fixedSize = 0; op.fixedCount = 0;
		vgx_device_scan(op, (Sum3*)a.partial, s, a.scan_bound); // (a frame's few hundred meshes: one launch instead of three)
		hipLaunchKernelGGL(k_asm_cmd_finish, dim3(2048), dim3(256), 0, s, a);
	} else {
		hipLaunchKernelGGL(k_asm_assign, dim3(2048), dim3(256), 0, s, a);
	}
	if (a.uv && a.uv_bytes) {
		hipLaunchKernelGGL(k_asm_uv, dim3(4096), dim3(256), 0, s, a);
	}
}
