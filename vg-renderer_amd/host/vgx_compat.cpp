// vgx_compat.cpp -- vg::pathXXX / vg::strokerXXX (include/vgx_compat.hpp) implemented on the C-ABI of libvgx.so.
// Host plumbing only: command recording, staging buffers, one vgx_* call sequence per API call. Geometry is
// computed by the HIP kernels; there is no CPU implementation here.
#include "../../include/vgx_compat.hpp"
#include "../../include/vgx.h"
#include <hip/hip_runtime_api.h>
#include <vector>
#include <string.h>

namespace vg
{
namespace {

int g_device = 0;

struct DevBuf
{
	void* p = nullptr;
	size_t cap = 0;
	bool ensure(size_t bytes)
	{
		if (bytes <= cap) { return true; }
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		const size_t want = bytes + bytes / 2 + 256;
		if (hipMalloc(&p, want) != hipSuccess) { return false; }
		cap = want;
		return true;
	}
	~DevBuf() { if (p) { (void)hipFree(p); } }
};

vgx_draw defaultDraw(float scale, float tol, float fringe)
{
	vgx_draw d;
	memset(&d, 0, sizeof(d));
	d.scale = scale; d.tess_tol = tol; d.fringe = fringe;
	d.mtx[0] = 1.0f; d.mtx[3] = 1.0f;
	return d;
}

} // namespace

// ---------------------------------------------------------------------------------------------------
struct Path
{
	vgx_ctx* ctx = nullptr;
	int status = VGX_OK;
	float scale = 1.0f, tol = 0.25f; // createPath defaults, reference path.cpp:28-29
	std::vector<uint8_t> types;
	std::vector<uint32_t> argOff;
	std::vector<float> args;
	// flattened result (lazy)
	bool dirty = true;
	std::vector<float> verts;
	std::vector<SubPath> subs;
	DevBuf dDraw, dPoly, dSubs;

	void cmd(uint8_t t, const float* a, uint32_t n)
	{
		types.push_back(t);
		args.insert(args.end(), a, a + n);
		argOff.push_back((uint32_t)args.size());
		dirty = true;
	}
	void flatten()
	{
		if (!dirty) { return; }
		dirty = false;
		verts.clear();
		subs.clear();
		status = VGX_OK;
		if (types.empty()) { return; }
		const uint32_t pcb[2] = { 0, (uint32_t)types.size() };
		vgx_pathset_desc desc;
		desc.cmd_type = types.data(); desc.cmd_arg_off = argOff.data(); desc.args = args.empty() ? nullptr : args.data();
		desc.path_cmd_begin = pcb; desc.npaths = 1; desc.ncmd = (uint32_t)types.size();
		static const float zero = 0.0f;
		if (!desc.args) { desc.args = &zero; }
		vgx_pathset* ps = nullptr;
		if ((status = vgx_pathset_create(ctx, &desc, &ps)) != VGX_OK) { return; }
		const vgx_draw d = defaultDraw(scale, tol, 1.0f);
		vgx_sizes sz;
		memset(&sz, 0, sizeof(sz));
		if (!dDraw.ensure(sizeof(d)) || hipMemcpy(dDraw.p, &d, sizeof(d), hipMemcpyHostToDevice) != hipSuccess) { status = VGX_E_HIP; }
		if (status == VGX_OK) { status = vgx_flatten_count(ctx, ps, (const vgx_draw*)dDraw.p, 1, &sz, nullptr); }
		if (status == VGX_OK && sz.num_poly_vertices) {
			std::vector<vgx_subpath> hs(sz.num_subpaths);
			if (!dPoly.ensure(sz.num_poly_vertices * 8) || !dSubs.ensure(sz.num_subpaths * sizeof(vgx_subpath))) { status = VGX_E_HIP; }
			vgx_flat_out out;
			out.poly = (float*)dPoly.p; out.subpaths = (vgx_subpath*)dSubs.p; out.draw_info = nullptr;
			out.cap_poly_vertices = sz.num_poly_vertices; out.cap_subpaths = sz.num_subpaths;
			if (status == VGX_OK) { status = vgx_flatten_emit(ctx, ps, (const vgx_draw*)dDraw.p, 1, 0, &out, nullptr); }
			if (status == VGX_OK) {
				verts.resize(sz.num_poly_vertices * 2);
				if (hipMemcpy(verts.data(), dPoly.p, sz.num_poly_vertices * 8, hipMemcpyDeviceToHost) != hipSuccess ||
				    hipMemcpy(hs.data(), dSubs.p, sz.num_subpaths * sizeof(vgx_subpath), hipMemcpyDeviceToHost) != hipSuccess) {
					status = VGX_E_HIP;
					verts.clear();
				} else {
					subs.resize(sz.num_subpaths);
					for (size_t i = 0; i < subs.size(); ++i) {
						subs[i].m_FirstVertexID = (uint32_t)hs[i].first_vertex;
						subs[i].m_NumVertices = hs[i].num_vertices;
						subs[i].m_IsClosed = (hs[i].flags & 1u) != 0;
					}
				}
			}
		}
		(void)vgx_pathset_destroy(ctx, ps);
	}
};

void vgxCompatSetDevice(int device) { g_device = device; }
int vgxCompatLastStatus(const Path* path) { return path ? path->status : VGX_E_INVALID_ARG; }

Path* createPath(bx::AllocatorI*)
{
	Path* p = new Path();
	if (vgx_create(g_device, &p->ctx) != VGX_OK) { delete p; return nullptr; }
	p->argOff.push_back(0);
	return p;
}

void destroyPath(Path* path)
{
	if (!path) { return; }
	vgx_ctx* c = path->ctx;
	delete path;
	(void)vgx_destroy(c);
}

void pathReset(Path* path, float scale, float tol) // path.cpp:44-60
{
	path->scale = scale; path->tol = tol;
	path->types.clear(); path->args.clear(); path->argOff.assign(1, 0u);
	path->dirty = true;
}

void pathMoveTo(Path* p, float x, float y) { const float a[] = { x, y }; p->cmd(VGX_CMD_MOVE_TO, a, 2); }
void pathLineTo(Path* p, float x, float y) { const float a[] = { x, y }; p->cmd(VGX_CMD_LINE_TO, a, 2); }
void pathCubicTo(Path* p, float c1x, float c1y, float c2x, float c2y, float x, float y) { const float a[] = { c1x, c1y, c2x, c2y, x, y }; p->cmd(VGX_CMD_CUBIC_TO, a, 6); }
void pathQuadraticTo(Path* p, float cx, float cy, float x, float y) { const float a[] = { cx, cy, x, y }; p->cmd(VGX_CMD_QUAD_TO, a, 4); }
void pathArcTo(Path* p, float x1, float y1, float x2, float y2, float r) { const float a[] = { x1, y1, x2, y2, r }; p->cmd(VGX_CMD_ARC_TO, a, 5); }
void pathRect(Path* p, float x, float y, float w, float h) { const float a[] = { x, y, w, h }; p->cmd(VGX_CMD_RECT, a, 4); }
void pathRoundedRect(Path* p, float x, float y, float w, float h, float r) { const float a[] = { x, y, w, h, r }; p->cmd(VGX_CMD_ROUNDED_RECT, a, 5); }
void pathRoundedRectVarying(Path* p, float x, float y, float w, float h, float rtl, float rtr, float rbr, float rbl) { const float a[] = { x, y, w, h, rtl, rtr, rbr, rbl }; p->cmd(VGX_CMD_ROUNDED_RECT_VARYING, a, 8); }
void pathCircle(Path* p, float x, float y, float r) { const float a[] = { x, y, r }; p->cmd(VGX_CMD_CIRCLE, a, 3); }
void pathEllipse(Path* p, float x, float y, float rx, float ry) { const float a[] = { x, y, rx, ry }; p->cmd(VGX_CMD_ELLIPSE, a, 4); }
void pathArc(Path* p, float x, float y, float r, float a0, float a1, Winding::Enum dir) { const float a[] = { x, y, r, a0, a1, dir == Winding::CW ? 1.0f : 0.0f }; p->cmd(VGX_CMD_ARC, a, 6); }
void pathPolyline(Path* p, const float* coords, uint32_t numPoints) { p->cmd(VGX_CMD_POLYLINE, coords, numPoints * 2); }
void pathClose(Path* p) { p->cmd(VGX_CMD_CLOSE, nullptr, 0); }

const float* pathGetVertices(const Path* path) { Path* p = const_cast<Path*>(path); p->flatten(); return p->verts.data(); }
uint32_t pathGetNumVertices(const Path* path) { Path* p = const_cast<Path*>(path); p->flatten(); return (uint32_t)(p->verts.size() / 2); }
const SubPath* pathGetSubPaths(const Path* path) { Path* p = const_cast<Path*>(path); p->flatten(); return p->subs.data(); }
uint32_t pathGetNumSubPaths(const Path* path) { Path* p = const_cast<Path*>(path); p->flatten(); return (uint32_t)p->subs.size(); }

// ---------------------------------------------------------------------------------------------------
struct Stroker
{
	vgx_ctx* ctx = nullptr;
	int status = VGX_OK;
	float scale = 1.0f, tol = 0.25f, fringe = 1.0f; // createStroker defaults, reference stroker.cpp:199-201
	std::vector<float> pos;
	std::vector<uint32_t> col;
	std::vector<uint16_t> idx;
	DevBuf dPoly, dSub, dSubDraw, dDraw, dPos, dCol, dIdx;

	// One strokerXXX call = one vertex list, one op.
	void run(Mesh* mesh, const float* vertexList, uint32_t n, bool closed, const vgx_draw& d, bool wantColor, bool aliasPos)
	{
		status = VGX_OK;
		vgx_subpath sp;
		sp.first_vertex = 0; sp.num_vertices = n; sp.flags = closed ? 1u : 0u;
		const uint32_t zero = 0;
		if (!dPoly.ensure((size_t)n * 8 + 8) || !dSub.ensure(sizeof(sp)) || !dSubDraw.ensure(4) || !dDraw.ensure(sizeof(d)) ||
		    hipMemcpy(dPoly.p, vertexList, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemcpy(dSub.p, &sp, sizeof(sp), hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemcpy(dSubDraw.p, &zero, 4, hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemcpy(dDraw.p, &d, sizeof(d), hipMemcpyHostToDevice) != hipSuccess) {
			status = VGX_E_HIP;
			return; // mesh untouched, like the reference's invalid-configuration path (stroker.cpp:269-271)
		}
		vgx_sizes sz;
		memset(&sz, 0, sizeof(sz));
		status = vgx_stroke_count(ctx, (const float*)dPoly.p, (const vgx_subpath*)dSub.p, (const uint32_t*)dSubDraw.p, 1, (const vgx_draw*)dDraw.p, 1, &sz, nullptr);
		if (status != VGX_OK || sz.num_meshes != 1) { return; }
		if (!dPos.ensure(sz.num_vertices * 8 + 16) || !dCol.ensure(sz.num_vertices * 4 + 16) || !dIdx.ensure(sz.num_indices * 2 + 16)) { status = VGX_E_HIP; return; }
		vgx_mesh_out out;
		out.pos = (float*)dPos.p; out.color = (uint32_t*)dCol.p; out.idx = (uint16_t*)dIdx.p; out.meshes = nullptr;
		out.cap_vertices = sz.num_vertices; out.cap_indices = sz.num_indices; out.cap_meshes = 0;
		status = vgx_stroke_emit(ctx, (const float*)dPoly.p, (const vgx_subpath*)dSub.p, (const uint32_t*)dSubDraw.p, 1, (const vgx_draw*)dDraw.p, 1, &out, nullptr);
		if (status != VGX_OK) { return; }
		pos.resize(sz.num_vertices * 2);
		col.resize(sz.num_vertices);
		idx.resize(sz.num_indices);
		if ((!aliasPos && hipMemcpy(pos.data(), dPos.p, sz.num_vertices * 8, hipMemcpyDeviceToHost) != hipSuccess) ||
		    (wantColor && hipMemcpy(col.data(), dCol.p, sz.num_vertices * 4, hipMemcpyDeviceToHost) != hipSuccess) ||
		    hipMemcpy(idx.data(), dIdx.p, sz.num_indices * 2, hipMemcpyDeviceToHost) != hipSuccess) {
			status = VGX_E_HIP;
			return;
		}
		mesh->m_PosBuffer = aliasPos ? vertexList : pos.data();
		mesh->m_ColorBuffer = wantColor ? col.data() : nullptr;
		mesh->m_IndexBuffer = idx.data();
		mesh->m_NumVertices = (uint32_t)sz.num_vertices;
		mesh->m_NumIndices = (uint32_t)sz.num_indices;
	}
};

int vgxCompatLastStatus(const Stroker* s) { return s ? s->status : VGX_E_INVALID_ARG; }

Stroker* createStroker(bx::AllocatorI*)
{
	Stroker* s = new Stroker();
	if (vgx_create(g_device, &s->ctx) != VGX_OK) { delete s; return nullptr; }
	return s;
}

void destroyStroker(Stroker* s)
{
	if (!s) { return; }
	vgx_ctx* c = s->ctx;
	delete s;
	(void)vgx_destroy(c);
}

void strokerReset(Stroker* s, float scale, float tol, float fringe) { s->scale = scale; s->tol = tol; s->fringe = fringe; } // stroker.cpp:232-237

static bool validCapJoin(uint32_t cap, uint32_t join) { return cap <= 2 && join <= 2; }

void strokerPolylineStroke(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, bool closed, float strokeWidth, LineCap::Enum cap, LineJoin::Enum join)
{
	if (!validCapJoin(cap, join) || n < 2) { return; } // invalid configuration: mesh left untouched (stroker.cpp:269-271)
	vgx_draw d = defaultDraw(s->scale, s->tol, s->fringe);
	d.stroke_flags = VGX_STROKE_FLAGS(cap, join, 0, 0);
	d.stroke_width = strokeWidth;
	s->run(mesh, vertexList, n, closed, d, false, false);
}

void strokerPolylineStrokeAA(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, bool closed, Color color, float strokeWidth, LineCap::Enum cap, LineJoin::Enum join)
{
	if (!validCapJoin(cap, join) || n < 2) { return; }
	vgx_draw d = defaultDraw(s->scale, s->tol, s->fringe);
	d.stroke_flags = VGX_STROKE_FLAGS(cap, join, 1, 0);
	d.stroke_width = strokeWidth;
	d.stroke_color = color;
	s->run(mesh, vertexList, n, closed, d, true, false);
}

void strokerPolylineStrokeAAThin(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, bool closed, Color color, LineCap::Enum cap, LineJoin::Enum join)
{
	if (!validCapJoin(cap, join) || n < 2) { return; }
	vgx_draw d = defaultDraw(s->scale, s->tol, s->fringe);
	d.stroke_flags = VGX_STROKE_FLAGS(cap, join, 1, 1);
	d.stroke_width = s->fringe;
	d.stroke_color = color;
	s->run(mesh, vertexList, n, closed, d, true, false);
}

void strokerConvexFill(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n)
{
	if (n < 3) { return; }
	vgx_draw d = defaultDraw(s->scale, s->tol, s->fringe);
	d.fill_flags = VGX_FILL_ENABLE;
	s->run(mesh, vertexList, n, false, d, false, true); // positions alias the caller's list (stroker.cpp:360)
}

void strokerConvexFillAA(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, uint32_t color)
{
	if (n < 3) { return; }
	vgx_draw d = defaultDraw(s->scale, s->tol, s->fringe);
	d.fill_flags = VGX_FILL_ENABLE | VGX_FILL_AA;
	d.fill_color = color;
	s->run(mesh, vertexList, n, false, d, true, false);
}
}
