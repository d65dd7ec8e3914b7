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
#include "vgx_internal.h"

namespace {

// first j in (i, M] with V(j) + nv(j) - V(i) > maxVB, where mesh M is a sentinel that never fits
__global__ __launch_bounds__(256) void k_asm_next(VgxAsmArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	const uint64_t M = A.totals->sizes.num_meshes;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= M; i += (uint64_t)gridDim.x * blockDim.x) {
		if (i == M) { A.jump0[M] = (uint32_t)M; continue; }
		const uint64_t limit = A.mtab[i].first_vertex + (uint64_t)A.max_vb;
		// meshes i..j-1 fit iff end(j-1) = V[j-1] + nv[j-1] <= limit; ends are non-decreasing
		uint64_t lo = i + 1, hi = M; // answer in [i+1, M]: mesh i itself always goes in (the reference VG_CHECKs nv < maxVB)
		while (lo < hi) {
			const uint64_t mid = (lo + hi) >> 1;
			const vgx_mesh m = A.mtab[mid];
			if (m.first_vertex + m.num_vertices <= limit) { lo = mid + 1; } else { hi = mid; }
		}
		A.jump0[i] = (uint32_t)lo;
	}
}

// one doubling round: n = number of buffer starts known so far (a power of two), J = next^n
__global__ __launch_bounds__(256) void k_asm_round(VgxAsmArgs A, const uint32_t* J, uint32_t* Jout, uint32_t n)
{
	if (A.totals->status != VGX_OK) { return; }
	const uint64_t M = A.totals->sizes.num_meshes;
	if (M == 0 || A.start[n - 1] >= M) { return; } // the chain already ran off the end: nothing left to extend
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= M; i += (uint64_t)gridDim.x * blockDim.x) {
		if (i < n) {
			const uint32_t s = A.start[i];
			A.start[n + i] = s >= M ? (uint32_t)M : J[s];
		}
		Jout[i] = J[J[i]];
	}
}

__global__ __launch_bounds__(256) void k_asm_init(VgxAsmArgs A)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < A.cap_start) { A.start[i] = i == 0 ? 0u : 0xFFFFFFFFu; }
}

__global__ __launch_bounds__(256) void k_asm_assign(VgxAsmArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	const uint64_t M = A.totals->sizes.num_meshes;
	// number of vertex buffers T = entries of start[] below M (start[] is increasing until it saturates at M / unset)
	uint64_t lo = 0, hi = A.cap_start;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if ((uint64_t)A.start[mid] < M) { lo = mid + 1; } else { hi = mid; }
	}
	const uint64_t T = lo;
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (tid == 0) {
		A.totals->sizes.num_drawcmds = T;
		if (T > A.cap_drawcmds) { atomicCAS(&A.totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_NOSPACE); }
		if (A.dev_num_drawcmds) { *A.dev_num_drawcmds = T; }
	}
	if (T > A.cap_drawcmds) { return; }
	for (uint64_t i = tid; i < M; i += (uint64_t)gridDim.x * blockDim.x) {
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
			A.drawcmds[a] = c;
		}
	}
}

} // namespace

void vgx_launch_assemble(const VgxAsmArgs& a, hipStream_t s)
{
	hipLaunchKernelGGL(k_asm_init, dim3((unsigned)((a.cap_start + 255) / 256)), dim3(256), 0, s, a);
	hipLaunchKernelGGL(k_asm_next, dim3(2048), dim3(256), 0, s, a);
	const uint32_t* J = a.jump0;
	uint32_t* Jn = a.jump1;
	for (uint32_t n = 1; n < a.cap_start; n <<= 1) { // doubles the known prefix of start[] every round
		hipLaunchKernelGGL(k_asm_round, dim3(2048), dim3(256), 0, s, a, J, Jn, n);
		const uint32_t* t = J; J = Jn; Jn = (uint32_t*)t;
	}
	hipLaunchKernelGGL(k_asm_assign, dim3(2048), dim3(256), 0, s, a);
}
