// vgx_compat.cpp -- vg::pathXXX / vg::strokerXXX (include/vgx_compat.hpp): the reference's per-call API (include/vg/path.h:19-38,
// include/vg/stroker.h:11-85) over the product. Two ways to serve a call ("implement them on the host -- one GPU
// launch per strokerXXX call would be ~10 us for ~1 us of work"):
//   host   the product's lane code (csrc/vgx_pathsim.h, vgx_elem.h, vgx_concave_lane.h -- what the kernels run per lane)
//          executed element after element by host/vgx_host_backend.hip: ~1 us per call, no device needed;
//   device one vgx_* call sequence on the C-ABI of libvgx.so per API call (staging copies + HIP kernels).
// VGX_COMPAT_BACKEND=host|device|auto (or vgxCompatSetBackend) picks; auto = host, and the device for vertex lists of at least
// VGX_COMPAT_DEVICE_MIN vertices (default 16384). Both produce the same bits (tests/compat_test.cpp runs every call on both).
// This file is plumbing: command recording, arrays from the caller's allocator, the libtess2 hand-over. Nothing under oracle/.
#include "../../include/vgx_compat.hpp"
#include "../../include/vgx.h"
#include "vgx_host_backend.h"
#include <hip/hip_runtime_api.h>
#include <vector>
#include <new>
#include <string.h>
#include <stdlib.h>

// bx's allocator interface (bx/allocator.h: a virtual destructor and ONE virtual function; alloc = realloc(nullptr, size),
// free = realloc(ptr, 0)) -- the reference's createPath / createStroker take it (path.cpp:23-30, stroker.cpp:194-200) and
// allocate the object and its growable arrays through it. bx itself is not vendored by the reference, so the interface is
// restated here; a caller that has bx passes its own allocator object, whose vtable has this layout.
namespace bx
{
struct AllocatorI
{
	virtual ~AllocatorI() = 0;
	virtual void* realloc(void* ptr, size_t size, size_t align, const char* file, uint32_t line) = 0;
};
}

namespace vg
{
namespace {

void* hostAlloc(bx::AllocatorI* a, size_t bytes)
{
	void* p = a ? a->realloc(nullptr, bytes, 0, __FILE__, __LINE__) : malloc(bytes);
	if (!p) { throw std::bad_alloc(); }
	return p;
}
void hostFree(bx::AllocatorI* a, void* p)
{
	if (!p) { return; }
	if (a) { (void)a->realloc(p, 0, 0, __FILE__, __LINE__); } else { free(p); }
}
// std allocator over the caller's bx allocator (nullptr: the C heap): every host array of a Path / Stroker comes from it
template<class T>
struct BxAlloc
{
	typedef T value_type;
	bx::AllocatorI* a;
	BxAlloc(bx::AllocatorI* alloc = nullptr) : a(alloc) {}
	template<class U> BxAlloc(const BxAlloc<U>& o) : a(o.a) {}
	T* allocate(size_t n) { return (T*)hostAlloc(a, n * sizeof(T)); }
	void deallocate(T* p, size_t) { hostFree(a, p); }
	template<class U> bool operator==(const BxAlloc<U>& o) const { return a == o.a; }
	template<class U> bool operator!=(const BxAlloc<U>& o) const { return a != o.a; }
};
template<class T> using Vec = std::vector<T, BxAlloc<T>>;

int g_device = 0;
enum { kBackendAuto = 0, kBackendHost = 1, kBackendDevice = 2 };
int g_backend = -1;          // -1: not read from the environment yet
uint32_t g_deviceMin = 16384;
bool g_envRead = false;
int backend()
{
	if (!g_envRead) { // the environment once; vgxCompatSetBackend overrides the backend, not the threshold
		g_envRead = true;
		const char* m = getenv("VGX_COMPAT_DEVICE_MIN");
		if (m && *m) { g_deviceMin = (uint32_t)strtoul(m, nullptr, 10); }
		if (g_backend < 0) {
			const char* e = getenv("VGX_COMPAT_BACKEND");
			g_backend = (e && !strcmp(e, "host")) ? kBackendHost : ((e && !strcmp(e, "device")) ? kBackendDevice : kBackendAuto);
		}
	}
	return g_backend < 0 ? (int)kBackendAuto : g_backend;
}
bool onDevice(uint32_t numVertices) { const int b = backend(); return b == kBackendDevice || (b == kBackendAuto && numVertices >= g_deviceMin); }
void* bxRealloc(void* user, void* ptr, size_t bytes) // vgxh::ReallocFn over the caller's bx::AllocatorI (nullptr: the C heap)
{
	bx::AllocatorI* a = (bx::AllocatorI*)user;
	if (a) { return a->realloc(ptr, bytes, 0, __FILE__, __LINE__); }
	if (!bytes) { free(ptr); return nullptr; }
	return realloc(ptr, bytes);
}
VgxTessApi g_tess = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
bool g_hasTess = false;
// libtess2's public constants (src/libtess2/tesselator.h:41-48, 110-115)
enum { kTessWindingOdd = 0, kTessWindingNonZero = 1, kTessPolygons = 0, kTessBoundaryContours = 2 };

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
	bx::AllocatorI* alloc;
	Vec<uint8_t> types;
	Vec<uint32_t> argOff;
	Vec<float> args;
	vgxh::Path* host = nullptr; // host backend: the builder executes every command as it arrives (like the reference)
	bool useDevice = false;     // fixed at creation / reset: a path is built by one backend from its first command on
	bool open = false, bad = false; // host backend: a sub-path can take more vertices / the path broke the grammar (empty until reset)
	uint32_t ncmd = 0;
	// device backend: flattened result (lazy)
	bool dirty = true;
	Vec<float> verts;
	Vec<SubPath> subs;
	DevBuf dDraw, dPoly, dSubs;
	explicit Path(bx::AllocatorI* a) : alloc(a), types(BxAlloc<uint8_t>(a)), argOff(BxAlloc<uint32_t>(a)), args(BxAlloc<float>(a)), verts(BxAlloc<float>(a)), subs(BxAlloc<SubPath>(a)) {}

	bool ensureCtx()
	{
		if (ctx) { return true; }
		status = vgx_create(g_device, &ctx);
		return status == VGX_OK;
	}
	void cmd(uint8_t t, const float* a, uint32_t n)
	{
		if (!useDevice) {
			// the command grammar of vgx_pathset_create (include/vgx.h; what the reference leaves undefined -- lineTo before
			// moveTo, appending to a closed sub-path, non-finite operands -- ends as a status and an empty path on both backends)
			if (bad) { return; }
			int st = VGX_OK;
			for (uint32_t i = 0; i < n; ++i) { if (!(a[i] - a[i] == 0.0f)) { st = VGX_E_NONFINITE; } }
			const bool isShape = t >= VGX_CMD_RECT && t <= VGX_CMD_ELLIPSE;
			if (st == VGX_OK) {
				if (t == VGX_CMD_ARC) {
					if (!(a[3] <= 1.0e5f && a[3] >= -1.0e5f && a[4] <= 1.0e5f && a[4] >= -1.0e5f)) { st = VGX_E_INVALID_ARG; }
					else if (!open && ncmd != 0) { st = VGX_E_INVALID_PATH; }
				} else if (t != VGX_CMD_MOVE_TO && !isShape && !open) {
					st = VGX_E_INVALID_PATH;
				} else if (t == VGX_CMD_POLYLINE && n < 2) {
					st = VGX_E_INVALID_ARG;
				}
			}
			if (st != VGX_OK) { status = st; bad = true; vgxh::pathReset(host, scale, tol); return; }
			open = !(t == VGX_CMD_CLOSE || isShape);
			++ncmd;
			if (!vgxh::pathCommand(host, t, a, n)) { status = VGX_E_HIP; bad = true; vgxh::pathReset(host, scale, tol); }
			return;
		}
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
		if (!ensureCtx()) { return; }
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
			Vec<vgx_subpath> hs(sz.num_subpaths, vgx_subpath(), BxAlloc<vgx_subpath>(alloc));
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
void vgxCompatSetBackend(int b) { g_backend = (b == kBackendHost || b == kBackendDevice) ? b : kBackendAuto; (void)backend(); }
int vgxCompatLastStatus(const Path* path) { return path ? path->status : VGX_E_INVALID_ARG; }

Path* createPath(bx::AllocatorI* allocator) // path.cpp:23-30: the object and its arrays come from `allocator`
{
	Path* p = ::new (hostAlloc(allocator, sizeof(Path))) Path(allocator);
	p->useDevice = backend() == kBackendDevice;
	p->host = vgxh::pathCreate(bxRealloc, allocator);
	if (!p->host || (p->useDevice && !p->ensureCtx())) { destroyPath(p); return nullptr; } // forced device backend without a device: loud
	p->argOff.push_back(0);
	return p;
}

void destroyPath(Path* path)
{
	if (!path) { return; }
	vgx_ctx* c = path->ctx;
	bx::AllocatorI* a = path->alloc;
	vgxh::pathDestroy(path->host);
	path->~Path();
	hostFree(a, path);
	if (c) { (void)vgx_destroy(c); }
}

void pathReset(Path* path, float scale, float tol) // path.cpp:44-60
{
	path->scale = scale; path->tol = tol;
	path->types.clear(); path->args.clear(); path->argOff.assign(1, 0u);
	path->dirty = true;
	path->status = VGX_OK;
	path->useDevice = backend() == kBackendDevice;
	path->open = false; path->bad = false; path->ncmd = 0;
	vgxh::pathReset(path->host, scale, tol);
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

static_assert(sizeof(SubPath) == sizeof(vgxh::SubRec) && offsetof(SubPath, m_NumVertices) == offsetof(vgxh::SubRec, n) && offsetof(SubPath, m_IsClosed) == offsetof(vgxh::SubRec, closed), "SubPath layout");
const float* pathGetVertices(const Path* path) { Path* p = const_cast<Path*>(path); if (!p->useDevice) { return vgxh::pathVertices(p->host); } p->flatten(); return p->verts.data(); }
uint32_t pathGetNumVertices(const Path* path) { Path* p = const_cast<Path*>(path); if (!p->useDevice) { return vgxh::pathNumVertices(p->host); } p->flatten(); return (uint32_t)(p->verts.size() / 2); }
const SubPath* pathGetSubPaths(const Path* path) { Path* p = const_cast<Path*>(path); if (!p->useDevice) { return (const SubPath*)vgxh::pathSubPaths(p->host); } p->flatten(); return p->subs.data(); }
uint32_t pathGetNumSubPaths(const Path* path) { Path* p = const_cast<Path*>(path); if (!p->useDevice) { return vgxh::pathNumSubPaths(p->host); } p->flatten(); return (uint32_t)p->subs.size(); }

// ---------------------------------------------------------------------------------------------------
struct Stroker
{
	vgx_ctx* ctx = nullptr;
	int status = VGX_OK;
	float scale = 1.0f, tol = 0.25f, fringe = 1.0f; // createStroker defaults, reference stroker.cpp:199-201
	bx::AllocatorI* alloc;
	Vec<float> pos;
	Vec<uint32_t> col;
	Vec<uint16_t> idx;
	DevBuf dPoly, dSub, dSubDraw, dDraw, dPos, dCol, dIdx;
	void* tess = nullptr;                    // strokerConcaveFill*: the host's libtess2 object
	void (*tessDelete)(void*) = nullptr;     // ... and the deleteTess of the table that made it (the table may be replaced / removed while the object lives)
	DevBuf dContour, dCont, dFill, dMoved, dTessPos, dTessIdx;
	Vec<float> moved, contourCopy;
	explicit Stroker(bx::AllocatorI* a) : alloc(a), pos(BxAlloc<float>(a)), col(BxAlloc<uint32_t>(a)), idx(BxAlloc<uint16_t>(a)), moved(BxAlloc<float>(a)), contourCopy(BxAlloc<float>(a)) {}

	bool ensureCtx()
	{
		if (ctx) { return true; }
		status = vgx_create(g_device, &ctx);
		return status == VGX_OK;
	}
	// host backend: the element code of the kernels, one element after the other (host/vgx_host_backend.hip)
	void runHost(Mesh* mesh, const float* vertexList, uint32_t n, bool closed, const vgx_draw& d, uint32_t kind, bool wantColor, bool aliasPos)
	{
		uint32_t nv = 0, ni = 0;
		status = vgxh::meshSize(vertexList, n, closed, &d, kind, &nv, &ni);
		if (status != VGX_OK) { return; } // mesh untouched, like the reference's invalid-configuration path (stroker.cpp:269-271)
		pos.resize((size_t)nv * 2 + 4); col.resize((size_t)nv + 2); idx.resize((size_t)ni + 8); // slack: the element code stores pairs / triples
		vgxh::meshEmit(vertexList, n, closed, &d, kind, pos.data(), col.data(), idx.data());
		mesh->m_PosBuffer = aliasPos ? vertexList : pos.data();
		mesh->m_ColorBuffer = wantColor ? col.data() : nullptr;
		mesh->m_IndexBuffer = idx.data();
		mesh->m_NumVertices = nv;
		mesh->m_NumIndices = ni;
	}
	// One strokerXXX call = one vertex list, one op.
	void run(Mesh* mesh, const float* vertexList, uint32_t n, bool closed, const vgx_draw& d, uint32_t kind, bool wantColor, bool aliasPos)
	{
		status = VGX_OK;
		if (!onDevice(n)) { runHost(mesh, vertexList, n, closed, d, kind, wantColor, aliasPos); return; }
		if (!ensureCtx()) {
			// auto backend on a box without a usable GPU: the host lane code serves the call (same arithmetic); only a FORCED
			// device backend fails loudly (mesh untouched, vgxCompatLastStatus says why)
			if (backend() == kBackendAuto) { status = VGX_OK; runHost(mesh, vertexList, n, closed, d, kind, wantColor, aliasPos); }
			return;
		}
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

Stroker* createStroker(bx::AllocatorI* allocator) // stroker.cpp:194-200
{
	Stroker* s = ::new (hostAlloc(allocator, sizeof(Stroker))) Stroker(allocator);
	if (backend() == kBackendDevice && !s->ensureCtx()) { s->~Stroker(); hostFree(allocator, s); return nullptr; } // forced device backend without a device: loud
	return s;
}

void destroyStroker(Stroker* s)
{
	if (!s) { return; }
	if (s->tess && s->tessDelete) { s->tessDelete(s->tess); }
	vgx_ctx* c = s->ctx;
	bx::AllocatorI* a = s->alloc;
	s->~Stroker();
	hostFree(a, s);
	if (c) { (void)vgx_destroy(c); }
}

void strokerReset(Stroker* s, float scale, float tol, float fringe) { s->scale = scale; s->tol = tol; s->fringe = fringe; } // stroker.cpp:232-237

static bool validCapJoin(uint32_t cap, uint32_t join) { return cap <= 2 && join <= 2; }

void strokerPolylineStroke(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, bool closed, float strokeWidth, LineCap::Enum cap, LineJoin::Enum join)
{
	if (!validCapJoin(cap, join) || n < 2) { return; } // invalid configuration: mesh left untouched (stroker.cpp:269-271)
	vgx_draw d = defaultDraw(s->scale, s->tol, s->fringe);
	d.stroke_flags = VGX_STROKE_FLAGS(cap, join, 0, 0);
	d.stroke_width = strokeWidth;
	s->run(mesh, vertexList, n, closed, d, VGX_MESH_STROKE, false, false);
}

void strokerPolylineStrokeAA(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, bool closed, Color color, float strokeWidth, LineCap::Enum cap, LineJoin::Enum join)
{
	if (!validCapJoin(cap, join) || n < 2) { return; }
	vgx_draw d = defaultDraw(s->scale, s->tol, s->fringe);
	d.stroke_flags = VGX_STROKE_FLAGS(cap, join, 1, 0);
	d.stroke_width = strokeWidth;
	d.stroke_color = color;
	s->run(mesh, vertexList, n, closed, d, VGX_MESH_STROKE_AA, true, false);
}

void strokerPolylineStrokeAAThin(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, bool closed, Color color, LineCap::Enum cap, LineJoin::Enum join)
{
	if (!validCapJoin(cap, join) || n < 2) { return; }
	vgx_draw d = defaultDraw(s->scale, s->tol, s->fringe);
	d.stroke_flags = VGX_STROKE_FLAGS(cap, join, 1, 1);
	d.stroke_width = s->fringe;
	d.stroke_color = color;
	s->run(mesh, vertexList, n, closed, d, VGX_MESH_STROKE_AA_THIN, true, false);
}

void strokerConvexFill(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n)
{
	if (n < 3) { return; }
	vgx_draw d = defaultDraw(s->scale, s->tol, s->fringe);
	d.fill_flags = VGX_FILL_ENABLE;
	s->run(mesh, vertexList, n, false, d, VGX_MESH_FILL, false, true); // positions alias the caller's list (stroker.cpp:360)
}

void strokerConvexFillAA(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, uint32_t color)
{
	if (n < 3) { return; }
	vgx_draw d = defaultDraw(s->scale, s->tol, s->fringe);
	d.fill_flags = VGX_FILL_ENABLE | VGX_FILL_AA;
	d.fill_color = color;
	s->run(mesh, vertexList, n, false, d, VGX_MESH_FILL_AA, true, false);
}

// ---- concave fills (include/vg/stroker.h:73-85, src/stroker.cpp:809-1006) ---------------------------------------------
void vgxCompatSetTessellator(const VgxTessApi* api)
{
	g_hasTess = api && api->newTess && api->deleteTess && api->addContour && api->tesselate && api->getVertexCount && api->getVertices && api->getElementCount && api->getElements;
	if (g_hasTess) { g_tess = *api; }
}

bool strokerConcaveFillBegin(Stroker* s) // stroker.cpp:809-845 (the scratch allocator is the host library's business)
{
	if (s->tess && s->tessDelete) { s->tessDelete(s->tess); }
	s->tess = nullptr; s->tessDelete = nullptr;
	if (!g_hasTess) { s->status = VGX_E_INVALID_ARG; return false; }
	s->tess = g_tess.newTess(nullptr);
	s->tessDelete = g_tess.deleteTess;
	return s->tess != nullptr;
}

void strokerConcaveFillAddContour(Stroker* s, const float* vertexList, uint32_t numVertices) // stroker.cpp:847-850
{
	if (s->tess && g_hasTess) { g_tess.addContour(s->tess, 2, vertexList, (int)(sizeof(float) * 2), (int)numVertices); }
}

bool strokerConcaveFillEnd(Stroker* s, Mesh* mesh, FillRule::Enum fillRule) // stroker.cpp:852-866: libtess2 only
{
	if (!s->tess || !g_hasTess) { return false; }
	if (!g_tess.tesselate(s->tess, fillRule == FillRule::NonZero ? kTessWindingNonZero : kTessWindingOdd, kTessPolygons, 3, 2, nullptr)) { return false; }
	mesh->m_PosBuffer = g_tess.getVertices(s->tess);
	mesh->m_ColorBuffer = nullptr;
	mesh->m_IndexBuffer = g_tess.getElements(s->tess);
	mesh->m_NumVertices = (uint32_t)g_tess.getVertexCount(s->tess);
	mesh->m_NumIndices = (uint32_t)g_tess.getElementCount(s->tess) * 3;
	return true;
}

bool strokerConcaveFillEndAA(Stroker* s, Mesh* mesh, uint32_t color, FillRule::Enum fillRule) // stroker.cpp:868-1006
{
	if (!s->tess || !g_hasTess) { return false; }
	const int rule = fillRule == FillRule::NonZero ? kTessWindingNonZero : kTessWindingOdd;
	const float normal[3] = { 0.0f, 0.0f, 1.0f };
	s->status = VGX_OK;
	// (1) boundary contours: libtess2, CPU
	if (!g_tess.tesselate(s->tess, rule, kTessBoundaryContours, 1, 2, normal)) { return false; }
	const float* contourVerts = g_tess.getVertices(s->tess);
	const unsigned short* contourData = g_tess.getElements(s->tess);
	const int numContours = g_tess.getElementCount(s->tess);
	const uint32_t numContourVerts = (uint32_t)g_tess.getVertexCount(s->tess);
	Vec<vgx_contour> contours((size_t)numContours, vgx_contour(), BxAlloc<vgx_contour>(s->alloc));
	uint32_t used = 0;
	for (int i = 0; i < numContours; ++i) {
		contours[i].first_vertex = contourData[2 * i]; contours[i].num_vertices = contourData[2 * i + 1]; contours[i].fill = 0;
		used += contourData[2 * i + 1];
	}
	vgx_concave_fill fill;
	memset(&fill, 0, sizeof(fill));
	fill.first_contour = 0; fill.num_contours = (uint32_t)numContours; fill.color = color; fill.fringe = s->fringe;
	// (2) moved contours
	s->moved.assign((size_t)numContourVerts * 2, 0.0f);
	bool dev = onDevice(numContourVerts);
	if (dev && !s->ensureCtx()) {
		if (backend() != kBackendAuto) { return false; } // forced device backend without a device: loud
		dev = false; s->status = VGX_OK;                 // auto: the host lane code serves the call
	}
	if (!dev) { // host backend: the same per-vertex function the kernels run (csrc/vgx_concave_lane.h)
		s->contourCopy.assign(contourVerts, contourVerts + (size_t)numContourVerts * 2); // the tessellator's arrays die in step (3)
		vgxh::concaveMove(s->contourCopy.data(), contours.data(), (uint32_t)numContours, s->fringe, s->moved.data());
	} else if (numContours > 0) {
		if (!s->dContour.ensure((size_t)numContourVerts * 8) || !s->dMoved.ensure((size_t)numContourVerts * 8) || !s->dCont.ensure(contours.size() * sizeof(vgx_contour)) || !s->dFill.ensure(sizeof(fill)) ||
		    hipMemcpy(s->dContour.p, contourVerts, (size_t)numContourVerts * 8, hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemcpy(s->dCont.p, contours.data(), contours.size() * sizeof(vgx_contour), hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemcpy(s->dFill.p, &fill, sizeof(fill), hipMemcpyHostToDevice) != hipSuccess) { s->status = VGX_E_HIP; return false; }
		s->status = vgx_concave_move(s->ctx, (const float*)s->dContour.p, numContourVerts, (const vgx_contour*)s->dCont.p, (uint64_t)numContours, (const vgx_concave_fill*)s->dFill.p, 1, (float*)s->dMoved.p, nullptr);
		if (s->status != VGX_OK || hipMemcpy(s->moved.data(), s->dMoved.p, (size_t)numContourVerts * 8, hipMemcpyDeviceToHost) != hipSuccess) { if (s->status == VGX_OK) { s->status = VGX_E_HIP; } return false; }
	}
	// (3) polygons of the moved contours: libtess2, CPU (the contour table above was copied: the tessellator's arrays die here)
	for (int i = 0; i < numContours; ++i) {
		g_tess.addContour(s->tess, 2, &s->moved[(size_t)contours[i].first_vertex * 2], (int)(sizeof(float) * 2), (int)contours[i].num_vertices);
	}
	if (!g_tess.tesselate(s->tess, rule, kTessPolygons, 3, 2, normal)) { return false; }
	fill.num_tess_vertices = (uint32_t)g_tess.getVertexCount(s->tess);
	fill.num_tess_indices = (uint32_t)g_tess.getElementCount(s->tess) * 3;
	const uint32_t nv = 2 * used + fill.num_tess_vertices, ni = 6 * used + fill.num_tess_indices;
	if (!dev) {
		if (nv > 65536u) { s->status = VGX_E_MESH_TOO_LARGE; return false; }
		s->pos.resize((size_t)nv * 2); s->col.resize(nv); s->idx.resize(ni);
		vgxh::concaveEmit(s->contourCopy.data(), contours.data(), (uint32_t)numContours, s->fringe, color, g_tess.getVertices(s->tess), fill.num_tess_vertices,
		                  g_tess.getElements(s->tess), fill.num_tess_indices, s->pos.data(), s->col.data(), s->idx.data());
		mesh->m_PosBuffer = s->pos.data();
		mesh->m_ColorBuffer = s->col.data();
		mesh->m_IndexBuffer = s->idx.data();
		mesh->m_NumVertices = nv;
		mesh->m_NumIndices = ni;
		return true;
	}
	// (2) + (4) the mesh: device
	if (!s->dTessPos.ensure((size_t)fill.num_tess_vertices * 8 + 8) || !s->dTessIdx.ensure((size_t)fill.num_tess_indices * 2 + 8) ||
	    !s->dPos.ensure((size_t)nv * 8 + 16) || !s->dCol.ensure((size_t)nv * 4 + 16) || !s->dIdx.ensure((size_t)ni * 2 + 16) || !s->dFill.ensure(sizeof(fill)) ||
	    (fill.num_tess_vertices && hipMemcpy(s->dTessPos.p, g_tess.getVertices(s->tess), (size_t)fill.num_tess_vertices * 8, hipMemcpyHostToDevice) != hipSuccess) ||
	    (fill.num_tess_indices && hipMemcpy(s->dTessIdx.p, g_tess.getElements(s->tess), (size_t)fill.num_tess_indices * 2, hipMemcpyHostToDevice) != hipSuccess) ||
	    hipMemcpy(s->dFill.p, &fill, sizeof(fill), hipMemcpyHostToDevice) != hipSuccess) { s->status = VGX_E_HIP; return false; }
	vgx_mesh_out out;
	out.pos = (float*)s->dPos.p; out.color = (uint32_t*)s->dCol.p; out.idx = (uint16_t*)s->dIdx.p; out.meshes = nullptr;
	out.cap_vertices = nv; out.cap_indices = ni; out.cap_meshes = 0;
	s->status = vgx_concave_emit(s->ctx, (const float*)s->dContour.p, numContourVerts, (const vgx_contour*)s->dCont.p, (uint64_t)numContours, (const vgx_concave_fill*)s->dFill.p, 1,
	                             (const float*)s->dTessPos.p, (const uint16_t*)s->dTessIdx.p, &out, nullptr, nullptr, nullptr);
	if (s->status != VGX_OK) { return false; }
	s->pos.resize((size_t)nv * 2); s->col.resize(nv); s->idx.resize(ni);
	if ((nv && (hipMemcpy(s->pos.data(), s->dPos.p, (size_t)nv * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(s->col.data(), s->dCol.p, (size_t)nv * 4, hipMemcpyDeviceToHost) != hipSuccess)) ||
	    (ni && hipMemcpy(s->idx.data(), s->dIdx.p, (size_t)ni * 2, hipMemcpyDeviceToHost) != hipSuccess)) { s->status = VGX_E_HIP; return false; }
	mesh->m_PosBuffer = s->pos.data();
	mesh->m_ColorBuffer = s->col.data();
	mesh->m_IndexBuffer = s->idx.data();
	mesh->m_NumVertices = nv;
	mesh->m_NumIndices = ni;
	return true;
}
}
