// vgx_merge.hip -- two mesh sequences of one frame merged into one, in draw order (vgx_merge, include/vgx.h).
//
// Concave fills (reference ctxFillPath*, src/vg.cpp:3133-3178, 3245-3277: strokerConcaveFillBegin / AddContour / End[AA]) are
// triangulated by libtess2, which stays on the CPU with the caller; their meshes are built OUTSIDE vgx_tessellate
// (vgx_flatten_* for the contours, libtess2, vgx_concave_move / vgx_concave_emit). The reference appends every mesh to the
// frame in submission order (createDrawCommand_VertexColor, vg.cpp:5207-5244), so the frame's final streams are the two
// sequences -- A: what vgx_tessellate made of the frame's draws (a concave draw contributes nothing there), B: the external
// meshes, each tagged with the frame draw it belongs to -- interleaved by draw index. Both are sorted by draw already:
//   rank of A[i] = i + #{B meshes with draw <  A[i].draw}      (two binary searches, no sort)
//   rank of B[j] = j + #{A meshes with draw <= B[j].draw}
// then one scan over the merged sequence gives the offsets, and one wave per mesh copies it (indices + the mesh's base inside
// its vertex buffer when draw-command assembly is armed: the merged frame is assembled exactly like a tessellated one).
#include "vgx_internal.h"
#include "vgx_wave.h"
#include "vgx_scan.h"
#include "vgx_scan_ops.h"

namespace {

__device__ __forceinline__ uint32_t merge_b_draw(const VgxMergeArgs& A, uint64_t j) { return A.b_draw ? A.b_draw[j] : A.b.meshes[j].draw; }

__global__ __launch_bounds__(256) void k_merge_rank(VgxMergeArgs A)
{
	const uint64_t na = A.a.num_meshes, nb = A.b.num_meshes;
	for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < na + nb; k += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t rank;
		if (k < na) {
			const uint32_t d = A.a.meshes[k].draw;
			uint64_t lo = 0, hi = nb; // first B with draw >= d
			while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (merge_b_draw(A, mid) < d) { lo = mid + 1; } else { hi = mid; } }
			rank = k + lo;
			if (k > 0 && A.a.meshes[k - 1].draw > d) { set_status(A.totals, VGX_E_INVALID_ARG); } // not sorted by draw
		} else {
			const uint64_t j = k - na;
			const uint32_t d = merge_b_draw(A, j);
			uint64_t lo = 0, hi = na; // first A with draw > d
			while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (A.a.meshes[mid].draw <= d) { lo = mid + 1; } else { hi = mid; } }
			rank = j + lo;
			if (j > 0 && merge_b_draw(A, j - 1) > d) { set_status(A.totals, VGX_E_INVALID_ARG); }
		}
		A.order[rank] = (uint32_t)(k < na ? k : (k - na) | 0x80000000u);
	}
}

struct OpMerge // scan over the merged sequence: offsets + the merged mesh table
{
	VgxMergeArgs A;
	__device__ uint64_t size() const { return A.totals->status == VGX_OK ? A.a.num_meshes + A.b.num_meshes : 0; }
	__device__ const vgx_mesh* src(uint64_t k) const
	{
		const uint32_t o = A.order[k];
		return (o >> 31) ? A.b.meshes + (o & 0x7FFFFFFFu) : A.a.meshes + o;
	}
	__device__ Sum3 load(uint64_t k) const
	{
		Sum3 r = sum3_zero();
		const vgx_mesh* m = src(k);
		r.a = m->num_vertices; r.b = m->num_indices;
		if (m->num_vertices > 65536u) { set_status(A.totals, VGX_E_MESH_TOO_LARGE); }
		return r;
	}
	__device__ void store(uint64_t k, Sum3 e) const
	{
		const uint32_t o = A.order[k];
		vgx_mesh m = *src(k);
		if (o >> 31) { m.draw = merge_b_draw(A, o & 0x7FFFFFFFu); }
		m.first_vertex = e.a; m.first_index = e.b;
		A.mtab[k] = m;
		A.mdesc[k].draw = m.draw; // draw-command assembly reads the mesh's draw (state key) from the descriptor
		if (A.meshes_out && k < A.caps.meshes) { A.meshes_out[k] = m; }
	}
	__device__ void finish(Sum3 t) const
	{
		const uint64_t n = A.a.num_meshes + A.b.num_meshes;
		A.totals->sizes.num_meshes = n;
		A.totals->sizes.num_vertices = t.a;
		A.totals->sizes.num_indices = t.b;
		if (t.a > A.caps.vertices || t.b > A.caps.indices || (A.meshes_out && n > A.caps.meshes)) { set_status(A.totals, VGX_E_NOSPACE); }
	}
};

__global__ __launch_bounds__(256) void k_merge_copy(VgxMergeArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	const int lane = threadIdx.x & 63;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	const uint64_t n = A.a.num_meshes + A.b.num_meshes;
	for (uint64_t k = wave; k < n; k += nwaves) {
		const uint32_t o = A.order[k];
		const bool fromB = (o >> 31) != 0;
		const vgx_cache_desc& S = fromB ? A.b : A.a;
		const vgx_mesh sm = S.meshes[o & 0x7FFFFFFFu];
		const vgx_mesh dm = A.mtab[k];
		const uint16_t base = A.mesh_base ? (uint16_t)A.mesh_base[k] : (uint16_t)0;
		for (uint32_t i = lane; i < sm.num_vertices; i += 64) {
			*(float2*)(A.pos + 2 * (dm.first_vertex + i)) = *(const float2*)(S.pos + 2 * (sm.first_vertex + i));
			A.color[dm.first_vertex + i] = S.color[sm.first_vertex + i];
		}
		for (uint32_t i = lane; i < sm.num_indices; i += 64) {
			A.idx[dm.first_index + i] = (uint16_t)(S.idx[sm.first_index + i] + base); // uint16 wrap = the reference's cast, vg_util.cpp:447
		}
		if (fromB && A.b_uv && A.uv_out) { // user meshes with texture coordinates: over the white-pixel UV the assembly step wrote
			const uint32_t words = A.uv_bytes / 4;
			const uint32_t* src = (const uint32_t*)A.b_uv + (size_t)sm.first_vertex * words;
			uint32_t* dst = (uint32_t*)A.uv_out + (size_t)dm.first_vertex * words;
			for (uint32_t i = lane; i < sm.num_vertices * words; i += 64) { dst[i] = src[i]; }
		}
	}
}

} // namespace

void vgx_launch_merge_rank(const VgxMergeArgs& a, hipStream_t s)
{
	const uint64_t n = a.a.num_meshes + a.b.num_meshes;
	const uint64_t g = (n + 255) / 256;
	if (n) { hipLaunchKernelGGL(k_merge_rank, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, a); }
}

void vgx_launch_merge_scan(const VgxMergeArgs& a, void* partial, hipStream_t s)
{
	OpMerge op; op.A = a;
	vgx_device_scan(op, (Sum3*)partial, s, a.a.num_meshes + a.b.num_meshes);
}

void vgx_launch_merge_copy(const VgxMergeArgs& a, hipStream_t s)
{
	const uint64_t n = a.a.num_meshes + a.b.num_meshes;
	const uint64_t g = (n + 3) / 4; // four waves per workgroup, one mesh per wave
	if (n) { hipLaunchKernelGGL(k_merge_copy, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, s, a); }
}
