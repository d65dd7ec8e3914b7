// vgx_cache.hip -- shape-cache instancing on gfx950.
//
// The reference's own answer to tessellation cost is its shape cache: the meshes a fill / stroke call produced are
// kept in the LOCAL space of the drawing (addCachedCommand, src/vg.cpp:5808-5841: positions times the inverse of the
// state transform, vgutil::invertMatrix3 + batchTransformPositions) and every later submission only transforms them
// with the current state transform and copies colours and indices into the frame's buffers (submitCachedMesh,
// vg.cpp:6137-6166 -> batchTransformPositions + createDrawCommand_VertexColor). Here:
//   k_cache_localize  one wave per mesh: pos <- inverse(draw.mtx) * pos                      (record time)
//   scan over the instances (vertices / indices / meshes of each instance's mesh range)      (submit time)
//   k_cache_meshes    one lane per (instance, mesh) pair: the output mesh table
//   k_cache_copy_flat instance-wise streaming copy: transformed positions + colours, and indices
//   k_cache_copy_idx_mesh  (assembly armed) indices per output mesh, plus its index base
// It is a pure streaming path: 21 B written per vertex, the cached drawing stays in L2.
#include "vgx_internal.h"
#include "vgx_wave.h"

namespace {

// vgutil::invertMatrix3 (vg_util.cpp:14-33): double-precision determinant and products, results rounded to float
__device__ __forceinline__ void invert_matrix3(const float* t, float* inv)
{
	const double det = (double)t[0] * t[3] - (double)t[2] * t[1];
	if (det > -1e-6 && det < 1e-6) {
		inv[0] = 1.0f; inv[2] = 1.0f; // sic: the reference sets inv[0] = inv[2] = 1 (vg_util.cpp:19)
		inv[1] = 0.0f; inv[3] = 0.0f; inv[4] = 0.0f; inv[5] = 0.0f;
		return;
	}
	const double invdet = 1.0 / det;
	inv[0] = (float)(t[3] * invdet);
	inv[2] = (float)(-t[2] * invdet);
	inv[4] = (float)(((double)t[2] * t[5] - (double)t[3] * t[4]) * invdet);
	inv[1] = (float)(-t[1] * invdet);
	inv[3] = (float)(t[0] * invdet);
	inv[5] = (float)(((double)t[1] * t[4] - (double)t[0] * t[5]) * invdet);
}

__global__ __launch_bounds__(VGX_WAVE) void k_cache_localize(const vgx_draw* draws, uint64_t ndraws, float* pos, const vgx_mesh* meshes, uint64_t numMeshes)
{
	const int lane = threadIdx.x;
	for (uint64_t m = blockIdx.x; m < numMeshes; m += gridDim.x) {
		const vgx_mesh me = meshes[m];
		if (me.draw >= ndraws) { continue; }
		float inv[6];
		invert_matrix3(draws[me.draw].mtx, inv);
		float2* p = (float2*)pos + me.first_vertex;
		for (uint32_t i = lane; i < me.num_vertices; i += VGX_WAVE) {
			const float2 q = p[i];
			const V2 r = v2xform(v2(q.x, q.y), inv); // transformPos2D, vg_util.h:24-28
			p[i] = make_float2(r.x, r.y);
		}
	}
}

__device__ __forceinline__ uint64_t cache_v(const VgxCacheArgs& A, uint64_t k) { return k < A.cache.num_meshes ? A.cache.meshes[k].first_vertex : A.cache.num_vertices; }
__device__ __forceinline__ uint64_t cache_i(const VgxCacheArgs& A, uint64_t k) { return k < A.cache.num_meshes ? A.cache.meshes[k].first_index : A.cache.num_indices; }

// one lane per output mesh = (instance, mesh of its range) pair
__global__ __launch_bounds__(256) void k_cache_meshes(VgxCacheArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	const uint64_t P = A.totals->sizes.num_meshes;
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t i = find_owner_u64(A.inst_mesh_prefix, 0, A.ninst, p); // last instance with prefix <= p ...
		// ... instances with an empty range share their successor's prefix: find_owner returns the LAST such entry,
		// which is the one that owns p
		const vgx_cache_instance in = A.inst[i];
		const uint64_t cm = in.first_mesh + (p - A.inst_mesh_prefix[i]);
		const vgx_mesh src = A.cache.meshes[cm];
		vgx_mesh r;
		r.first_vertex = A.inst_vert_prefix[i] + (src.first_vertex - cache_v(A, in.first_mesh));
		r.first_index = A.inst_idx_prefix[i] + (src.first_index - cache_i(A, in.first_mesh));
		r.num_vertices = src.num_vertices;
		r.num_indices = src.num_indices;
		r.draw = (uint32_t)i; // the instance takes the place of the draw
		r.subpath_kind = src.subpath_kind;
		A.mtab[p] = r;
		if (A.meshes_out) { A.meshes_out[p] = r; }
		{ const uint32_t kind = src.subpath_kind >> 28; if (kind == VGX_MESH_FILL || kind == VGX_MESH_STROKE) { A.totals->cache_has_uniform = 1u; } } // plain store, rare
	}
}

struct __attribute__((packed, aligned(2))) Idx4 { uint32_t a, b; };

// An instance's mesh range is ONE contiguous block of the cached vertex / index streams and lands in one contiguous
// block of the output, so positions, colours and (unless the assembly step needs a per-mesh base) indices are copied
// instance-wise: a wave owns a contiguous range of the OUTPUT stream and walks the instances that intersect it --
// full 64-lane accesses whatever the mesh sizes are, no dependent per-mesh record loads in the loop.
template<int WHAT> // 0: positions + colours (item = vertex), 1: indices (item = 4 indices)
__global__ __launch_bounds__(VGX_WAVE) void k_cache_copy_flat(VgxCacheArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	const int lane = threadIdx.x;
	const uint64_t* prefix = WHAT == 0 ? A.inst_vert_prefix : A.inst_idx_prefix;
	const uint64_t total = prefix[A.ninst];
	// ranges are multiples of 256 items so that the 4x-unrolled loop below mostly runs full
	uint64_t per = (total + gridDim.x - 1) / gridDim.x;
	per = (per + 255) / 256 * 256;
	const uint64_t o0 = (uint64_t)blockIdx.x * per;
	const uint64_t o1 = o0 + per < total ? o0 + per : total;
	if (o0 >= o1) { return; }
	uint64_t i = find_owner_u64(prefix, 0, A.ninst, o0); // instance that owns output item o0 (skips empty ranges)
	uint64_t o = o0;
	while (o < o1) {
		const uint64_t ib = prefix[i], ie = prefix[i + 1];
		if (ie <= o) { ++i; continue; }
		const vgx_cache_instance in = A.inst[i];
		const uint64_t end = ie < o1 ? ie : o1;
		if (WHAT == 0) {
			const uint64_t sbase = cache_v(A, in.first_mesh) + (o - ib);
			const float2* sp = (const float2*)A.cache.pos + sbase;
			const uint32_t* sc = A.cache.color + sbase;
			float2* dp = (float2*)A.pos + o;
			uint32_t* dc = A.color + o;
			const uint64_t n = end - o;
			for (uint64_t k = lane; k < n; k += VGX_WAVE) {
				const float2 q = sp[k];
				const V2 r = v2xform(v2(q.x, q.y), in.mtx); // batchTransformPositions, vg.cpp:6162
				dp[k] = make_float2(r.x, r.y);
				dc[k] = sc[k];
			}
		} else {
			const uint16_t* si = A.cache.idx + cache_i(A, in.first_mesh) + (o - ib);
			uint16_t* di = A.idx + o;
			const uint64_t n = end - o;
			const uint64_t n4 = n >> 2;
			for (uint64_t k = lane; k < n4; k += VGX_WAVE) {
				*(Idx4*)(di + 4 * k) = *(const Idx4*)(si + 4 * k);
			}
			for (uint64_t k = 4 * n4 + lane; k < n; k += VGX_WAVE) { di[k] = si[k]; }
		}
		o = end;
	}
}

// Meshes cached WITHOUT per-vertex colours (the non-AA flavours: CachedMesh::m_Colors == nullptr, addCachedCommand
// vg.cpp:5826-5834) are drawn with the colour of the command that replays them (submitCachedMesh :6159-6160), not with what
// the caching frame wrote -- the two differ for thin non-AA strokes, whose alpha ctxStrokePathColor scales only while
// caching. After the flat copy: one lane per output mesh, AA meshes are skipped at once.
__global__ __launch_bounds__(256) void k_cache_uniform_colors(VgxCacheArgs A)
{
	if (A.totals->status != VGX_OK || A.totals->cache_has_uniform == 0u) { return; } // nothing but AA meshes: all colours were stored
	const uint64_t P = A.totals->sizes.num_meshes;
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (uint64_t)gridDim.x * blockDim.x) {
		const vgx_mesh m = A.mtab[p];
		const uint32_t kind = m.subpath_kind >> 28;
		if (kind != VGX_MESH_FILL && kind != VGX_MESH_STROKE) { continue; }
		const uint32_t c = A.inst[m.draw].color; // draw = instance index (k_cache_meshes)
		uint32_t* dc = A.color + m.first_vertex;
		for (uint32_t v = 0; v < m.num_vertices; ++v) { dc[v] = c; }
	}
}

// Assembly armed: every mesh has its own index base -> one wave per output mesh for the index stream.
__global__ __launch_bounds__(VGX_WAVE) void k_cache_copy_idx_mesh(VgxCacheArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	const int lane = threadIdx.x;
	const uint64_t P = A.totals->sizes.num_meshes;
	const uint64_t per = (P + gridDim.x - 1) / gridDim.x;
	const uint64_t p0 = (uint64_t)blockIdx.x * per;
	const uint64_t p1 = p0 + per < P ? p0 + per : P;
	for (uint64_t p = p0; p < p1; ++p) {
		const vgx_mesh dst = A.mtab[p];
		const uint64_t i = dst.draw;
		const vgx_cache_instance in = A.inst[i];
		const uint64_t cm = in.first_mesh + (p - A.inst_mesh_prefix[i]);
		const vgx_mesh src = A.cache.meshes[cm];
		const uint32_t base = A.mesh_base[p]; // vertices in front of the mesh inside its vertex buffer
		const uint32_t base2 = (base & 0xFFFFu) * 0x10001u;
		const uint16_t* si = A.cache.idx + src.first_index;
		uint16_t* di = A.idx + dst.first_index;
		const uint32_t n4 = src.num_indices >> 2;
		for (uint32_t k = lane; k < n4; k += VGX_WAVE) { // four indices per lane: unaligned 8-byte loads / stores
			Idx4 v = *(const Idx4*)(si + 4 * k);
			// packed uint16 add without carry between the halves
			v.a = ((v.a & 0x7FFF7FFFu) + (base2 & 0x7FFF7FFFu)) ^ ((v.a ^ base2) & 0x80008000u);
			v.b = ((v.b & 0x7FFF7FFFu) + (base2 & 0x7FFF7FFFu)) ^ ((v.b ^ base2) & 0x80008000u);
			*(Idx4*)(di + 4 * k) = v;
		}
		for (uint32_t k = 4 * n4 + lane; k < src.num_indices; k += VGX_WAVE) {
			di[k] = (uint16_t)(si[k] + base);
		}
	}
}

} // namespace

void vgx_launch_cache_localize(const vgx_draw* draws, uint64_t ndraws, float* pos, const vgx_mesh* meshes, uint64_t numMeshes, hipStream_t s)
{
	const uint64_t g = numMeshes < 32768 ? (numMeshes ? numMeshes : 1) : 32768;
	hipLaunchKernelGGL(k_cache_localize, dim3((unsigned)g), dim3(VGX_WAVE), 0, s, draws, ndraws, pos, meshes, numMeshes);
}

void vgx_launch_cache_meshes(const VgxCacheArgs& a, hipStream_t s)
{
	hipLaunchKernelGGL(k_cache_meshes, dim3(2048), dim3(256), 0, s, a);
}

void vgx_launch_cache_copy(const VgxCacheArgs& a, int numBlocks, hipStream_t s)
{
	hipLaunchKernelGGL(k_cache_copy_flat<0>, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
	hipLaunchKernelGGL(k_cache_uniform_colors, dim3(2048), dim3(256), 0, s, a);
	if (a.mesh_base) {
		hipLaunchKernelGGL(k_cache_copy_idx_mesh, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
	} else {
		hipLaunchKernelGGL(k_cache_copy_flat<1>, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
	}
}
