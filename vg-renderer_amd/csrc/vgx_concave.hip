// vgx_concave.hip -- the stroker's own arithmetic inside strokerConcaveFillEndAA (reference src/stroker.cpp:868-1006)
// for a batch of concave fills; libtess2 stays with the caller on the CPU.
//
// The reference alternates libtess2 and its own loops:
//   (1) tessTesselate(TESS_BOUNDARY_CONTOURS) of the added contours                              [caller, CPU]
//   (2) per contour vertex: two fringe vertices {p[inner], colour} {p[1 - inner], colour & 0x00FFFFFF}, six indices per
//       contour segment, and the contour vertex MOVES to p[inner] (stroker.cpp:887-973)           [k_concave_move, k_concave_fringe]
//   (3) tessAddContour of the moved contours + tessTesselate(TESS_POLYGONS)                       [caller, CPU]
//   (4) interior appended behind the fringe: positions copied, colour replicated, indices rebased by the fringe's
//       vertex count with vgutil::batchTransformDrawIndices (stroker.cpp:976-994)                  [k_concave_interior]
// One lane = one contour vertex (2) or one interior vertex / index (4).
//
// Quirk reproduced: the loop of (2) updates the contour in place, so the LAST vertex of a contour takes its outgoing
// direction towards the ALREADY MOVED vertex 0 (stroker.cpp:900-901, 920); every other vertex sees original neighbours.
#include "vgx_internal.h"
#include "vgx_wave.h"
#include "vgx_concave_lane.h"

namespace {

// owner contour of flat contour-vertex index e: last contour with first_vertex <= e (contours are stored back to back)
__device__ __forceinline__ uint64_t contour_of(const vgx_contour* c, uint64_t n, uint64_t e)
{
	uint64_t lo = 0, hi = n;
	while (hi - lo > 1) {
		const uint64_t mid = (lo + hi) >> 1;
		if (c[mid].first_vertex <= e) { lo = mid; } else { hi = mid; }
	}
	return lo;
}

__global__ __launch_bounds__(256) void k_concave_move(VgxConcaveArgs A)
{
	for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < A.num_contour_vertices; e += (uint64_t)gridDim.x * blockDim.x) {
		const vgx_contour c = A.contours[contour_of(A.contours, A.ncontours, e)];
		const uint32_t j = (uint32_t)(e - c.first_vertex);
		if (j >= c.num_vertices || c.fill >= A.nfills) { continue; }
		const FringePair f = contour_vertex(A.contour_verts + 2 * c.first_vertex, c.num_vertices, j, A.fills[c.fill].fringe);
		*(float2*)(A.moved + 2 * e) = make_float2(f.in.x, f.in.y); // "Update contour vertex", stroker.cpp:917
	}
}

// per fill: vertex / index counts -> offsets + mesh record (scan operator, see vgx_scan.h); defined in vgx_api.hip

// The contour table is device memory the host never sees: before anything is written through it, every contour must point
// back at the fill whose range holds it, lie inside the vertex array and follow its predecessor (the fringe kernel derives a
// contour's place in its mesh from first_vertex differences). A bad table ends as VGX_E_INVALID_ARG, never as a stray write.
__global__ __launch_bounds__(256) void k_concave_validate(VgxConcaveArgs A)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.ncontours; i += (uint64_t)gridDim.x * blockDim.x) {
		const vgx_contour c = A.contours[i];
		bool ok = c.fill < A.nfills && c.first_vertex <= A.num_contour_vertices && c.num_vertices <= A.num_contour_vertices - c.first_vertex;
		if (ok) {
			const vgx_concave_fill fl = A.fills[c.fill];
			ok = i >= fl.first_contour && i - fl.first_contour < fl.num_contours;
		}
		if (ok && i > 0) {
			const vgx_contour q = A.contours[i - 1];
			ok = q.first_vertex <= c.first_vertex && q.num_vertices <= c.first_vertex - q.first_vertex;
		}
		if (!ok) { atomicCAS(&A.totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_INVALID_ARG); }
	}
}

__global__ __launch_bounds__(256) void k_concave_fringe(VgxConcaveArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < A.num_contour_vertices; e += (uint64_t)gridDim.x * blockDim.x) {
		const vgx_contour c = A.contours[contour_of(A.contours, A.ncontours, e)];
		const uint32_t j = (uint32_t)(e - c.first_vertex);
		if (j >= c.num_vertices) { continue; }
		const vgx_concave_fill fl = A.fills[c.fill];
		const vgx_mesh m = A.mtab[c.fill];
		const uint32_t n = c.num_vertices;
		const FringePair f = contour_vertex(A.contour_verts + 2 * c.first_vertex, n, j, fl.fringe);
		// nextVertexID / nextIndexID of this contour inside its mesh: two vertices and six indices per contour vertex in front
		const uint32_t before = (uint32_t)(c.first_vertex - A.contours[fl.first_contour].first_vertex);
		const uint32_t vb = 2 * before, ib = 6 * before;
		const uint64_t gv = m.first_vertex + vb + 2 * (uint64_t)j;
		*(float4*)(A.pos + 2 * gv) = make_float4(f.in.x, f.in.y, f.out.x, f.out.y);
		*(uint2*)(A.color + gv) = make_uint2(fl.color, fl.color & 0x00FFFFFFu); // colorSetAlpha(color, 0)
		// segment j: (id0, id2, id1) (id2, id3, id1); the closing segment wraps to the contour's first pair (stroker.cpp:934-967)
		const uint32_t id0 = vb + 2 * j, id1 = id0 + 1;
		const uint32_t id2 = (j + 1 < n) ? id0 + 2 : vb, id3 = id2 + 1;
		uint16_t* pi = A.idx + m.first_index + ib + 6 * (uint64_t)j;
		pi[0] = (uint16_t)id0; pi[1] = (uint16_t)id2; pi[2] = (uint16_t)id1;
		pi[3] = (uint16_t)id2; pi[4] = (uint16_t)id3; pi[5] = (uint16_t)id1;
	}
}

__global__ __launch_bounds__(256) void k_concave_interior(VgxConcaveArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	// one wave per fill walks its interior: vertices (copy + colour), then indices (rebase by the fringe's vertex count)
	const int lane = threadIdx.x & 63;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	for (uint64_t fi = wave; fi < A.nfills; fi += nwaves) {
		const vgx_concave_fill fl = A.fills[fi];
		const vgx_mesh m = A.mtab[fi];
		const uint32_t fringeV = m.num_vertices - fl.num_tess_vertices; // = nextVertexID after the last contour
		const uint32_t fringeI = m.num_indices - fl.num_tess_indices;
		for (uint32_t i = lane; i < fl.num_tess_vertices; i += 64) {
			const float2 p = *(const float2*)(A.tess_pos + 2 * (fl.first_tess_vertex + i));
			*(float2*)(A.pos + 2 * (m.first_vertex + fringeV + i)) = p;
			A.color[m.first_vertex + fringeV + i] = fl.color; // memset32, stroker.cpp:985
		}
		const uint16_t delta = (uint16_t)fringeV; // batchTransformDrawIndices(src, n, dst, (uint16_t)nextVertexID), stroker.cpp:992
		for (uint32_t i = lane; i < fl.num_tess_indices; i += 64) {
			A.idx[m.first_index + fringeI + i] = (uint16_t)(A.tess_idx[fl.first_tess_index + i] + delta);
		}
	}
}

} // namespace

void vgx_launch_concave_move(const VgxConcaveArgs& a, hipStream_t s)
{
	hipLaunchKernelGGL(k_concave_move, dim3(1024), dim3(256), 0, s, a);
}

void vgx_launch_concave_emit(const VgxConcaveArgs& a, hipStream_t s)
{
	hipLaunchKernelGGL(k_concave_validate, dim3(256), dim3(256), 0, s, a);
	hipLaunchKernelGGL(k_concave_fringe, dim3(1024), dim3(256), 0, s, a);
	hipLaunchKernelGGL(k_concave_interior, dim3(1024), dim3(256), 0, s, a);
}
