// vgx_host_backend.h -- host execution of the product's per-lane code for the per-call reference API (include/vgx_compat.hpp).
//
// the 32 vg::pathXXX / vg::strokerXXX symbols are per-call functions over ONE path or ONE polyline; a GPU launch per
// call costs ~10 us for ~1 us of work, so the boundary serves them on the host. What runs here is NOT the oracle and not a second
// implementation: it is the product's own lane code -- csrc/vgx_pathsim.h (vg::Path semantics), csrc/vgx_elem.h (one stroker element),
// csrc/vgx_concave_lane.h (one concave-fringe vertex), the very functions the HIP kernels execute one per lane -- compiled for the host
// and driven element after element instead of lane beside lane. The batch C-ABI (include/vgx.h: vgx_tessellate, vgx_flatten, ...)
// never comes here: it runs on the device or fails.
#ifndef VGX_HOST_BACKEND_H
#define VGX_HOST_BACKEND_H

#include <stdint.h>
#include <stddef.h>
#include "../../include/vgx.h"

namespace vgxh
{
// growable host arrays come from the caller (the compat layer hands its bx::AllocatorI through): realloc semantics,
// newBytes == 0 frees. Returns nullptr when out of memory.
typedef void* (*ReallocFn)(void* user, void* ptr, size_t newBytes);

struct SubRec { uint32_t first, n; bool closed; }; // layout of vg::SubPath (include/vg/path.h:11-16)

struct Path; // vg::Path state: the exact sequential builder + its arrays
Path* pathCreate(ReallocFn re, void* user);
void pathDestroy(Path* p);
void pathReset(Path* p, float scale, float tol);
// one path command (VGX_CMD_*), executed immediately like the reference's pathXXX. false: out of memory (path unchanged)
bool pathCommand(Path* p, uint32_t type, const float* args, uint32_t nargs);
const float* pathVertices(Path* p);
uint32_t pathNumVertices(Path* p);
const SubRec* pathSubPaths(Path* p); // includes the sub-path still being built
uint32_t pathNumSubPaths(Path* p);

// One stroker mesh from one vertex list. kind = VGX_MESH_FILL / FILL_AA / STROKE / STROKE_AA / STROKE_AA_THIN; `d` carries
// scale / tess_tol / fringe and the stroke / fill parameters exactly as a batch draw record does.
// meshSize: vertex / index counts (closed form, or one pass over the elements for Round joins); VGX_E_MESH_TOO_LARGE above 65536 vertices.
int meshSize(const float* poly, uint32_t n, bool closed, const vgx_draw* d, uint32_t kind, uint32_t* nv, uint32_t* ni);
// meshEmit: pos[nv][2], col[nv] (may be null for the non-AA kinds, whose colours nobody reads), idx[ni]
void meshEmit(const float* poly, uint32_t n, bool closed, const vgx_draw* d, uint32_t kind, float* pos, uint32_t* col, uint16_t* idx);

// strokerConcaveFillEndAA's own loops (stroker.cpp:887-994) around the caller's libtess2:
// moved[v] = inner fringe vertex of every contour vertex (what goes back into libtess2)
void concaveMove(const float* contourVerts, const vgx_contour* contours, uint32_t ncontours, float fringe, float* moved);
// the mesh: [2 vertices, 6 indices per contour vertex][interior, indices rebased]
void concaveEmit(const float* contourVerts, const vgx_contour* contours, uint32_t ncontours, float fringe, uint32_t color,
                 const float* tessPos, uint32_t numTessVerts, const uint16_t* tessIdx, uint32_t numTessIdx, float* pos, uint32_t* col, uint16_t* idx);
}

#endif
