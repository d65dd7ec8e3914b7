// vgx_compat.hpp -- the reference's own vg::pathXXX / vg::strokerXXX API on top of the C-ABI (include/vgx.h).
//
// Drop-in replacement for <vg/path.h> (reference include/vg/path.h:19-38) and <vg/stroker.h>
// (include/vg/stroker.h:11-72): same namespace, names, argument order and POD layouts, so the call sites in the
// reference's src/vg.cpp compile unchanged. A call is served in one of two ways (vgxCompatSetBackend below): by the
// product's own per-lane code compiled for the host (host/vgx_host_backend.hip: the functions the kernels run one per lane,
// driven element after element; ~1 us per call, no GPU needed -- the default, because one launch per strokerXXX call would cost
// ~10 us for ~1 us of work), or by the HIP kernels of libvgx.so with a batch of ONE path / ONE vertex list (a launch + copy
// round trip per call, ~0.1 ms; auto mode takes it for vertex lists of at least VGX_COMPAT_DEVICE_MIN vertices and falls back
// to the host code when no gfx950 device is usable; a FORCED device backend without a device fails loudly: the create
// functions return nullptr). Same bits either way. This layer exists for source compatibility and incremental adoption;
// throughput comes from batching through vgx.h. Nothing under oracle/ is behind it.
//
// createPath / createStroker honour the caller's bx::AllocatorI (bx/allocator.h's interface: a virtual destructor and realloc(ptr,
// size, align, file, line)): the object and every host-side array come from it, nullptr = the C heap; device buffers are hipMalloc.
//
// Differences from the reference, all documented in DESIGN.md:
//   - command grammar the reference leaves undefined (lineTo before moveTo, appending to a closed sub-path) yields an
//     empty path instead of undefined behaviour; NaN/Inf arguments likewise (they hang the reference, path.cpp:109);
//   - the bx transcendentals are the pinned ones of csrc/vgmath.h;
//   - strokerConcaveFill* need the host's libtess2 (the reference vendors it under src/libtess2 and links it into the same
//     binary): hand its eight entry points over once with vgxCompatSetTessellator(); libtess2 itself stays on the CPU, the
//     stroker's own fringe / rebase loops of strokerConcaveFillEndAA run on the device (vgx_concave_move / _emit).
#ifndef VGX_COMPAT_HPP
#define VGX_COMPAT_HPP

#include <stdint.h>

namespace bx { struct AllocatorI; }

namespace vg
{
#ifndef VG_H // the reference's include/vg/vg.h already defines these (identical layouts, vg.h:102,156-174,252-259,353-360)
typedef uint32_t Color;
struct LineCap { enum Enum : uint32_t { Butt = 0, Round = 1, Square = 2 }; };
struct LineJoin { enum Enum : uint32_t { Miter = 0, Round = 1, Bevel = 2 }; };
struct Winding { enum Enum : uint32_t { CCW = 0, CW = 1 }; };
struct FillRule { enum Enum : uint32_t { NonZero = 0, EvenOdd = 1 }; };
struct Mesh
{
	const float* m_PosBuffer;
	const uint32_t* m_ColorBuffer;
	const uint16_t* m_IndexBuffer;
	uint32_t m_NumVertices;
	uint32_t m_NumIndices;
};
#endif

#ifndef VG_PATH_H
struct SubPath // include/vg/path.h:11-16
{
	uint32_t m_FirstVertexID;
	uint32_t m_NumVertices;
	bool m_IsClosed;
};
#endif

struct Path;
struct Stroker;

// ---- include/vg/path.h:19-38 ----
Path* createPath(bx::AllocatorI* allocator);
void destroyPath(Path* path);
void pathReset(Path* path, float scale, float tesselationTolerance);
void pathMoveTo(Path* path, float x, float y);
void pathLineTo(Path* path, float x, float y);
void pathCubicTo(Path* path, float c1x, float c1y, float c2x, float c2y, float x, float y);
void pathQuadraticTo(Path* path, float cx, float cy, float x, float y);
void pathArcTo(Path* path, float x1, float y1, float x2, float y2, float r);
void pathRect(Path* path, float x, float y, float w, float h);
void pathRoundedRect(Path* path, float x, float y, float w, float h, float r);
void pathRoundedRectVarying(Path* path, float x, float y, float w, float h, float rtl, float rtr, float rbr, float rbl);
void pathCircle(Path* path, float x, float y, float r);
void pathEllipse(Path* path, float x, float y, float rx, float ry);
void pathArc(Path* path, float x, float y, float r, float a0, float a1, Winding::Enum dir);
void pathPolyline(Path* path, const float* coords, uint32_t numPoints);
void pathClose(Path* path);
const float* pathGetVertices(const Path* path);   // valid until the next mutation of `path`
uint32_t pathGetNumVertices(const Path* path);
const SubPath* pathGetSubPaths(const Path* path);
uint32_t pathGetNumSubPaths(const Path* path);

// ---- include/vg/stroker.h:11-72 ----
Stroker* createStroker(bx::AllocatorI* allocator);
void destroyStroker(Stroker* stroker);
void strokerReset(Stroker* stroker, float scale, float tesselationTolerance, float fringeWidth);
// Mesh pointers stay valid until the next strokerXXX call on the same Stroker (reference stroker.cpp:2316-2320).
void strokerPolylineStroke(Stroker* stroker, Mesh* mesh, const float* vertexList, uint32_t numVertices, bool isClosed, float strokeWidth, LineCap::Enum lineCap, LineJoin::Enum lineJoin);
void strokerPolylineStrokeAA(Stroker* stroker, Mesh* mesh, const float* vertexList, uint32_t numVertices, bool isClosed, Color color, float strokeWidth, LineCap::Enum lineCap, LineJoin::Enum lineJoin);
void strokerPolylineStrokeAAThin(Stroker* stroker, Mesh* mesh, const float* vertexList, uint32_t numVertices, bool isClosed, Color color, LineCap::Enum lineCap, LineJoin::Enum lineJoin);
void strokerConvexFill(Stroker* stroker, Mesh* mesh, const float* vertexList, uint32_t numVertices);
void strokerConvexFillAA(Stroker* stroker, Mesh* mesh, const float* vertexList, uint32_t numVertices, uint32_t color);

// include/vg/stroker.h:73-85. false (and *mesh untouched) when no tessellator was set or libtess2 fails.
bool strokerConcaveFillBegin(Stroker* stroker);
void strokerConcaveFillAddContour(Stroker* stroker, const float* vertexList, uint32_t numVertices);
bool strokerConcaveFillEnd(Stroker* stroker, Mesh* mesh, FillRule::Enum fillRule);
bool strokerConcaveFillEndAA(Stroker* stroker, Mesh* mesh, uint32_t color, FillRule::Enum fillRule);

// ---- not in the reference ----
// The host's libtess2: the eight functions of src/libtess2/tesselator.h that src/stroker.cpp:809-1006 calls, with their
// own signatures (TESStesselator* and TESSalloc* as void*). Typical binding:
//   static const vg::VgxTessApi api = { (void* (*)(void*))tessNewTess, (void (*)(void*))tessDeleteTess, (void (*)(void*, int, const void*, int, int))tessAddContour,
//       (int (*)(void*, int, int, int, int, const float*))tessTesselate, (int (*)(void*))tessGetVertexCount, (const float* (*)(void*))tessGetVertices,
//       (int (*)(void*))tessGetElementCount, (const unsigned short* (*)(void*))tessGetElements };
//   vg::vgxCompatSetTessellator(&api);
struct VgxTessApi
{
	void* (*newTess)(void* alloc);
	void (*deleteTess)(void* tess);
	void (*addContour)(void* tess, int size, const void* pointer, int stride, int count);
	int (*tesselate)(void* tess, int windingRule, int elementType, int polySize, int vertexSize, const float* normal);
	int (*getVertexCount)(void* tess);
	const float* (*getVertices)(void* tess);
	int (*getElementCount)(void* tess);
	const unsigned short* (*getElements)(void* tess);
};
void vgxCompatSetTessellator(const VgxTessApi* api); // copied; nullptr removes it. A Stroker's tessellator object is deleted (next Begin /
                                                     // destroyStroker) with the deleteTess of the table that made it: keep that libtess2 loaded while Strokers live
// Device used by subsequently created Path / Stroker objects (default 0). Last status of an object (vgx_status).
void vgxCompatSetDevice(int device);
// 0 = auto (host; the device for vertex lists of at least VGX_COMPAT_DEVICE_MIN vertices), 1 = host, 2 = device. Default: the
// environment variable VGX_COMPAT_BACKEND (host | device | auto). Host = the product's per-lane code executed on the CPU, ~1 us per
// call and no GPU needed; device = one C-ABI call sequence per call. Same bits either way.
void vgxCompatSetBackend(int backend);
int vgxCompatLastStatus(const Path* path);
int vgxCompatLastStatus(const Stroker* stroker);
}

#endif
