// vgo_port.h -- CPU restatement of vg-renderer's Path flattener and Stroker (TEST INFRASTRUCTURE).
//
// This is the "port" oracle: a from-scratch sequential C++ restatement of the algorithms in the
// reference's src/path.cpp and src/stroker.cpp (scalar build), written so that the repo still has a
// checker when /root/reference is absent (GPU box, CI). It is validated here against the reference's
// own compiled sources (oracle/_ref/libvgref.so, tests/test_oracle_vs_reference.py) and against the
// golden vectors generated from them (tests/golden/). Each function cites the reference lines it
// follows. Nothing in the product path may include, link or call this file.
//
// Parity status: pinned against the reference itself run in the build container; bx transcendentals
// are unpinned upstream and defined by csrc/vgmath.h (see that header).
#ifndef VGO_PORT_H
#define VGO_PORT_H

#include <stdint.h>
#include <bx/allocator.h>

namespace vgo
{
typedef uint32_t Color;

struct LineCap { enum Enum : uint32_t { Butt = 0, Round = 1, Square = 2 }; };
struct LineJoin { enum Enum : uint32_t { Miter = 0, Round = 1, Bevel = 2 }; };
struct Winding { enum Enum : uint32_t { CCW = 0, CW = 1 }; };

struct SubPath // include/vg/path.h:11-16
{
	uint32_t m_FirstVertexID;
	uint32_t m_NumVertices;
	bool m_IsClosed;
};

struct Mesh // include/vg/vg.h:353-360
{
	const float* m_PosBuffer;
	const uint32_t* m_ColorBuffer;
	const uint16_t* m_IndexBuffer;
	uint32_t m_NumVertices;
	uint32_t m_NumIndices;
};

struct Path;
struct Stroker;

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
const float* pathGetVertices(const Path* path);
uint32_t pathGetNumVertices(const Path* path);
const SubPath* pathGetSubPaths(const Path* path);
uint32_t pathGetNumSubPaths(const Path* path);

Stroker* createStroker(bx::AllocatorI* allocator);
void destroyStroker(Stroker* stroker);
void strokerReset(Stroker* stroker, float scale, float tesselationTolerance, float fringeWidth);
void strokerPolylineStroke(Stroker* stroker, Mesh* mesh, const float* vertexList, uint32_t numVertices, bool isClosed, float strokeWidth, LineCap::Enum lineCap, LineJoin::Enum lineJoin);
void strokerPolylineStrokeAA(Stroker* stroker, Mesh* mesh, const float* vertexList, uint32_t numVertices, bool isClosed, Color color, float strokeWidth, LineCap::Enum lineCap, LineJoin::Enum lineJoin);
void strokerPolylineStrokeAAThin(Stroker* stroker, Mesh* mesh, const float* vertexList, uint32_t numVertices, bool isClosed, Color color, LineCap::Enum lineCap, LineJoin::Enum lineJoin);
void strokerConvexFill(Stroker* stroker, Mesh* mesh, const float* vertexList, uint32_t numVertices);
void strokerConvexFillAA(Stroker* stroker, Mesh* mesh, const float* vertexList, uint32_t numVertices, uint32_t color);

// vgutil::batchTransformPositions, scalar form (src/vg_util.cpp:266-272, src/vg_util.h:24-28)
void batchTransformPositions(const float* v, uint32_t n, float* p, const float* mtx);
}

#endif
