// ref_capi.cpp -- C wrapper that drives the REFERENCE'S OWN compiled sources (oracle/_ref/libvgref.so).
// TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this. It is compiled together with /root/reference/src/{path,stroker,vg_util}.cpp (+libtess2,
// so stroker.cpp links) by oracle/Makefile; no reference source is copied into this repository.
// kind = "reference" in bench.py's cpu_baseline.
#include <vg/path.h>
#include <vg/stroker.h>
#include "vg_util.h" // /root/reference/src/vg_util.h
#include <bx/allocator.h>

#define VGO_ENGINE vg
#define VGO_ENGINE_NAME "reference(vg-renderer src @ /root/reference, scalar build, bx_shim+vgmath)"
#define VGO_XFORM vgutil::batchTransformPositions
#define VGO_REBASE(src, n, dst, delta) vgutil::batchTransformDrawIndices((src), (n), (dst), (delta)) // the reference's own (vg_util.cpp:447-520)
#define VGO_INVERT3(t, inv) vgutil::invertMatrix3((t), (inv))                      // vg_util.cpp:14-33
#include "vgo_driver.inl"

// ---- concave fills: only this oracle has them (libtess2 is linked into libvgref.so) ------------------------
// vgo_concave_fill_aa = the reference's complete strokerConcaveFillBegin / AddContour / EndAA for ONE fill (the checker).
// vgo_tess_* = "the caller's libtess2" of the tests: thin wrappers over the same library, driven in the order
// strokerConcaveFillEndAA drives it (ONE tesselator object: contours -> boundary contours -> moved contours -> polygons),
// so a test can do the CPU halves of the algorithm around the device calls vgx_concave_move / vgx_concave_emit.
#include "libtess2/tesselator.h"

extern "C" {

int vgo_concave_fill_aa(const float* verts, const uint32_t* contourFirst, const uint32_t* contourCount, uint32_t ncontours, uint32_t color, float fringe,
                        int evenOdd, float* pos, uint32_t* col, uint16_t* idx, uint32_t capV, uint32_t capI, uint32_t* nv, uint32_t* ni)
{
	bx::ShimAllocator alloc;
	vg::Stroker* stroker = vg::createStroker(&alloc);
	vg::strokerReset(stroker, 1.0f, 0.25f, fringe);
	vg::strokerConcaveFillBegin(stroker);
	for (uint32_t c = 0; c < ncontours; ++c) {
		vg::strokerConcaveFillAddContour(stroker, verts + 2 * contourFirst[c], contourCount[c]);
	}
	vg::Mesh mesh;
	const bool ok = vg::strokerConcaveFillEndAA(stroker, &mesh, color, evenOdd ? vg::FillRule::EvenOdd : vg::FillRule::NonZero);
	int rc = ok ? 0 : 1;
	if (ok) {
		*nv = mesh.m_NumVertices; *ni = mesh.m_NumIndices;
		if (mesh.m_NumVertices <= capV && mesh.m_NumIndices <= capI) {
			memcpy(pos, mesh.m_PosBuffer, sizeof(float) * 2 * mesh.m_NumVertices);
			memcpy(col, mesh.m_ColorBuffer, sizeof(uint32_t) * mesh.m_NumVertices);
			memcpy(idx, mesh.m_IndexBuffer, sizeof(uint16_t) * mesh.m_NumIndices);
		} else {
			rc = 2;
		}
	}
	vg::destroyStroker(stroker);
	return rc;
}

void* vgo_tess_new(void) { return tessNewTess(nullptr); }
void vgo_tess_delete(void* t) { tessDeleteTess((TESStesselator*)t); }
void vgo_tess_add_contour(void* t, const float* v, uint32_t n) { tessAddContour((TESStesselator*)t, 2, v, sizeof(float) * 2, (int)n); }
// boundary != 0: TESS_BOUNDARY_CONTOURS (polySize 1), else TESS_POLYGONS with triangles; the arguments of stroker.cpp:882, 976
int vgo_tess_run(void* t, int evenOdd, int boundary)
{
	const float normal[3] = { 0.0f, 0.0f, 1.0f };
	return tessTesselate((TESStesselator*)t, evenOdd ? TESS_WINDING_ODD : TESS_WINDING_NONZERO, boundary ? TESS_BOUNDARY_CONTOURS : TESS_POLYGONS, boundary ? 1 : 3, 2, normal);
}
// strokerConcaveFillEnd's call (stroker.cpp:852): triangles, NO normal given (libtess2 derives the projection itself)
int vgo_tess_run_plain(void* t, int evenOdd)
{
	return tessTesselate((TESStesselator*)t, evenOdd ? TESS_WINDING_ODD : TESS_WINDING_NONZERO, TESS_POLYGONS, 3, 2, nullptr);
}
int vgo_tess_vertex_count(void* t) { return tessGetVertexCount((TESStesselator*)t); }
const float* vgo_tess_vertices(void* t) { return tessGetVertices((TESStesselator*)t); }
int vgo_tess_element_count(void* t) { return tessGetElementCount((TESStesselator*)t); }
const uint16_t* vgo_tess_elements(void* t) { return tessGetElements((TESStesselator*)t); }

} // extern "C"
