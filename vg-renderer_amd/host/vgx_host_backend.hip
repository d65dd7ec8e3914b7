// vgx_host_backend.hip -- see vgx_host_backend.h. A HIP translation unit without kernels: it is compiled by hipcc so that the
// __host__ __device__ lane functions of csrc/ (the ones the kernels run) are instantiated for the host, with the same arithmetic
// flags as the device build (-ffp-contract=off -fno-fast-math; x86-64 SSE2 float + - * / sqrt are the same IEEE binary32 operations,
// transcendentals are csrc/vgmath.h's). Nothing under oracle/ is used or linked.
#include <hip/hip_runtime.h>
#include "vgx_host_backend.h"
#include "../csrc/vgx_pathsim.h"
#include "../csrc/vgx_elem.h"
#include "../csrc/vgx_concave_lane.h"
#include <string.h>
#include <math.h>

namespace vgxh
{
namespace {

struct HostStack // pending halves of the cubic walk (path.cpp:90: 10 levels), three points each (csrc/vgx_lane.h)
{
	float s[VGX_CUBIC_MAX_PENDING][6];
	void push(int level, float ax, float ay, float bx, float by, float cx, float cy)
	{
		float* p = s[level];
		p[0] = ax; p[1] = ay; p[2] = bx; p[3] = by; p[4] = cx; p[5] = cy;
	}
	void pop(int level, float& ax, float& ay, float& bx, float& by, float& cx, float& cy)
	{
		const float* p = s[level];
		ax = p[0]; ay = p[1]; bx = p[2]; by = p[3]; cx = p[4]; cy = p[5];
	}
};

struct VtxHost // vertex source of the element code: the caller's vertex list
{
	const float* p;
	V2 ld(uint32_t i) const { return v2(p[2 * (size_t)i], p[2 * (size_t)i + 1]); }
};
typedef MeshCtxT<VtxHost> HostMesh;

typedef PathSim<true, false> Sim;

// one command on the builder: PathSim::run's dispatch without the final endSub (the path stays open for the next call)
void step(Sim& sim, HostStack& st, uint32_t type, const float* a, uint32_t na)
{
	switch (type) {
	case VGX_CMD_MOVE_TO: sim.moveTo(a[0], a[1]); break;
	case VGX_CMD_LINE_TO: sim.lineTo(a[0], a[1]); break;
	case VGX_CMD_CUBIC_TO: sim.cubicTo(a[0], a[1], a[2], a[3], a[4], a[5], st); break;
	case VGX_CMD_QUAD_TO: sim.quadTo(a[0], a[1], a[2], a[3], st); break;
	case VGX_CMD_CLOSE: sim.close(); break;
	case VGX_CMD_ARC_TO: sim.arcTo(a[0], a[1], a[2], a[3], a[4]); break;
	case VGX_CMD_ARC: sim.arc(a[0], a[1], a[2], a[3], a[4], a[5] != 0.0f); break;
	case VGX_CMD_POLYLINE: sim.polyline(a, na >> 1); break;
	default: sim.shape(type, a); break;
	}
}

bool makeMesh(const float* poly, uint32_t n, bool closed, const vgx_draw* d, uint32_t kind, HostMesh* mc, VgxMeshPrep* pr, vgx_mesh* mt, VgxMeshDesc* md)
{
	// the descriptor / constants / closed-form size of one mesh, written by the same function the flatten kernels call
	const bool needsCount = vgx_write_mesh(md, mt, 0, d, 0, 0, kind, closed, 0, n, pr, poly);
	mc->kind = VGX_MD_KIND(md->kind);
	mc->closed = VGX_MD_CLOSED(md->kind) != 0;
	mc->cap = VGX_MD_CAP(md->kind);
	mc->join = VGX_MD_JOIN(md->kind);
	mc->N = n;
	mc->j = 0;
	mc->vtx.p = poly;
	mc->hsw = pr->f0; mc->hswAA = pr->f1; mc->fringe = pr->f2;
	mc->dr = d;
	return needsCount;
}

} // namespace

// ---- vg::Path -----------------------------------------------------------------------------------------------------------
struct Path
{
	ReallocFn re;
	void* user;
	Sim sim;
	HostStack st;
	float* verts;
	uint32_t cap;        // vertices
	vgx_subpath* subs;   // completed sub-paths as the builder writes them
	uint32_t subCap;
	SubRec* out;         // what pathSubPaths returns (rebuilt when asked after a change)
	uint32_t outCap;
	bool outDirty;
};

Path* pathCreate(ReallocFn re, void* user)
{
	Path* p = (Path*)re(user, nullptr, sizeof(Path));
	if (!p) { return nullptr; }
	memset(p, 0, sizeof(Path));
	p->re = re; p->user = user;
	pathReset(p, 1.0f, 0.25f); // createPath defaults, path.cpp:28-29
	return p;
}

void pathDestroy(Path* p)
{
	if (!p) { return; }
	ReallocFn re = p->re;
	void* user = p->user;
	if (p->verts) { re(user, p->verts, 0); }
	if (p->subs) { re(user, p->subs, 0); }
	if (p->out) { re(user, p->out, 0); }
	re(user, p, 0);
}

void pathReset(Path* p, float scale, float tol) // path.cpp:44-60
{
	Sim& s = p->sim;
	memset(&s, 0, sizeof(s));
	s.scale = scale; s.tol = tol;
	s.poly = p->verts; s.subs = p->subs; s.limit = p->cap;
	s.init();
	p->outDirty = true;
}

bool pathCommand(Path* p, uint32_t type, const float* args, uint32_t nargs)
{
	Sim& s = p->sim;
	if (p->subCap < s.nsubs + 2) { // a command completes at most one sub-path and opens at most one
		const uint32_t want = p->subCap ? 2 * p->subCap : 16;
		vgx_subpath* q = (vgx_subpath*)p->re(p->user, p->subs, (size_t)want * sizeof(vgx_subpath));
		if (!q) { return false; }
		p->subs = q; p->subCap = want; s.subs = q;
	}
	const Sim before = s;
	for (;;) {
		step(s, p->st, type, args, nargs);
		// the builder writes vertices [0, limit) only; pathClose may pop ONE vertex after the peak, so `peak <= nverts + 1`
		if ((uint64_t)s.nverts + 1 <= p->cap) { break; }
		uint64_t want = (uint64_t)p->cap * 2;
		if (want < (uint64_t)s.nverts + 65) { want = (uint64_t)s.nverts + 65; }
		if (want > 0x7FFFFFFFull) { s = before; return false; }
		float* q = (float*)p->re(p->user, p->verts, (size_t)want * 2 * sizeof(float));
		if (!q) { s = before; return false; }
		p->verts = q; p->cap = (uint32_t)want;
		s = before; // replay the command on the larger array: same arithmetic, same result
		s.poly = q; s.limit = p->cap;
	}
	p->outDirty = true;
	return true;
}

const float* pathVertices(Path* p) { return p->verts; }
uint32_t pathNumVertices(Path* p) { return p->sim.nverts; }
uint32_t pathNumSubPaths(Path* p) { return p->sim.nsubs; }

const SubRec* pathSubPaths(Path* p)
{
	const Sim& s = p->sim;
	if (!p->outDirty) { return p->out; }
	if (p->outCap < s.nsubs) {
		const uint32_t want = s.nsubs + 16;
		SubRec* q = (SubRec*)p->re(p->user, p->out, (size_t)want * sizeof(SubRec));
		if (!q) { return nullptr; }
		p->out = q; p->outCap = want;
	}
	const uint32_t done = s.open ? s.nsubs - 1 : s.nsubs;
	for (uint32_t i = 0; i < done; ++i) {
		p->out[i].first = (uint32_t)p->subs[i].first_vertex; p->out[i].n = p->subs[i].num_vertices; p->out[i].closed = (p->subs[i].flags & 1u) != 0;
	}
	if (s.open) { p->out[done].first = s.spFirst; p->out[done].n = s.spN; p->out[done].closed = s.spClosed; }
	p->outDirty = false;
	return p->out;
}

// ---- vg::Stroker: one mesh from one vertex list ------------------------------------------------------------------------------
int meshSize(const float* poly, uint32_t n, bool closed, const vgx_draw* d, uint32_t kind, uint32_t* nvOut, uint32_t* niOut)
{
	for (size_t i = 0; i < (size_t)n * 2; ++i) { if (!isfinite(poly[i])) { return VGX_E_NONFINITE; } }
	HostMesh mc; VgxMeshPrep pr; vgx_mesh mt; VgxMeshDesc md;
	const bool needsCount = makeMesh(poly, n, closed, d, kind, &mc, &pr, &mt, &md);
	uint64_t nv = mt.num_vertices, ni = mt.num_indices;
	if (needsCount) { // Round joins: numArcPoints per join is data dependent (stroker.cpp:1146, 1592)
		nv = 0; ni = 0;
		V2 dPrev = v2dir(mc.vtx.ld(n - 1), mc.vtx.ld(0));
		for (uint32_t j = 0; j < n; ++j) {
			mc.j = j;
			const V2 p1 = mc.vtx.ld(j);
			const V2 d12 = v2dir(p1, mc.vtx.ld(j + 1 < n ? j + 1 : 0));
			const Elem e = elem_geometry(mc, p1, dPrev, d12);
			nv += e.nv;
			ni += elem_total_indices(mc, e);
			dPrev = d12;
		}
	}
	if (nv > 65536u || ni > 0xFFFFFFFFull) { return VGX_E_MESH_TOO_LARGE; } // uint16 indices
	*nvOut = (uint32_t)nv; *niOut = (uint32_t)ni;
	return VGX_OK;
}

void meshEmit(const float* poly, uint32_t n, bool closed, const vgx_draw* d, uint32_t kind, float* pos, uint32_t* col, uint16_t* idx)
{
	HostMesh mc; VgxMeshPrep pr; vgx_mesh mt; VgxMeshDesc md;
	makeMesh(poly, n, closed, d, kind, &mc, &pr, &mt, &md);
	if (kind < VGX_MESH_STROKE) { // strokerConvexFill / strokerConvexFillAA: one polygon corner after the other (stroker.cpp:334-365, 713-807)
		FillFetch F;
		memset(&F, 0, sizeof(F));
		F.valid = true; F.aaElem = kind == VGX_MESH_FILL_AA; F.sseOrder = VGX_MD_SSE_ORDER(md.kind) != 0;
		F.N = n; F.color = pr.color; F.aa = pr.f0;
		V2 dPrev = v2(0.0f, 0.0f);
		if (F.aaElem) { dPrev = v2dir(mc.vtx.ld(n - 1), mc.vtx.ld(0)); }
		for (uint32_t j = 0; j < n; ++j) {
			F.j = j;
			F.p1 = mc.vtx.ld(j);
			V2 d12 = v2(0.0f, 0.0f);
			if (F.aaElem) { d12 = v2dir(F.p1, mc.vtx.ld(j + 1 < n ? j + 1 : 0)); }
			fill_emit_store(pos, col, idx, F, dPrev, d12);
			dPrev = d12;
		}
		return;
	}
	// polyline strokes (stroker.cpp:1008-2314): stroke_chunk's steps A-D, one element at a time -- the running vertex / index
	// bases and the previous element's exit rails are plain variables instead of wave scans
	StrokeWriter w;
	w.pos = pos; w.col = col; w.idx = idx;
	w.color = pr.color; w.c0 = pr.color & 0x00FFFFFFu; w.ib = 0;
	uint32_t vb = 0, ib = 0;
	Rails prev = rails(0, 0, 0, 0);
	V2 dPrev = v2dir(mc.vtx.ld(n - 1), mc.vtx.ld(0));
	for (uint32_t j = 0; j < n; ++j) {
		mc.j = j;
		const V2 p1 = mc.vtx.ld(j);
		const V2 d12 = v2dir(p1, mc.vtx.ld(j + 1 < n ? j + 1 : 0));
		const Elem e = elem_geometry(mc, p1, dPrev, d12);
		const uint32_t total = elem_total_indices(mc, e);
		w.reset();
		elem_emit(mc, e, vb, ib, prev, w);
		w.flush(vb, ib);
		prev = elem_exit_rails(mc, e, vb);
		vb += e.nv; ib += total;
		dPrev = d12;
	}
}

// ---- strokerConcaveFillEndAA's loops -------------------------------------------------------------------------------------
void concaveMove(const float* contourVerts, const vgx_contour* contours, uint32_t ncontours, float fringe, float* moved)
{
	for (uint32_t c = 0; c < ncontours; ++c) {
		const float* v = contourVerts + 2 * contours[c].first_vertex;
		const uint32_t n = contours[c].num_vertices;
		for (uint32_t j = 0; j < n; ++j) {
			const FringePair f = contour_vertex(v, n, j, fringe);
			float* o = moved + 2 * (contours[c].first_vertex + j);
			o[0] = f.in.x; o[1] = f.in.y; // "Update contour vertex", stroker.cpp:917
		}
	}
}

void concaveEmit(const float* contourVerts, const vgx_contour* contours, uint32_t ncontours, float fringe, uint32_t color,
                 const float* tessPos, uint32_t numTessVerts, const uint16_t* tessIdx, uint32_t numTessIdx, float* pos, uint32_t* col, uint16_t* idx)
{
	uint32_t vb = 0, ib = 0; // nextVertexID / nextIndexID (stroker.cpp:884-885)
	for (uint32_t c = 0; c < ncontours; ++c) {
		const float* v = contourVerts + 2 * contours[c].first_vertex;
		const uint32_t n = contours[c].num_vertices;
		for (uint32_t j = 0; j < n; ++j) {
			const FringePair f = contour_vertex(v, n, j, fringe);
			float* o = pos + 2 * (size_t)(vb + 2 * j);
			o[0] = f.in.x; o[1] = f.in.y; o[2] = f.out.x; o[3] = f.out.y;
			col[vb + 2 * j] = color; col[vb + 2 * j + 1] = color & 0x00FFFFFFu; // colorSetAlpha(color, 0)
			const uint32_t id0 = vb + 2 * j, id1 = id0 + 1;
			const uint32_t id2 = (j + 1 < n) ? id0 + 2 : vb, id3 = id2 + 1; // the closing segment wraps (stroker.cpp:934-967)
			uint16_t* pi = idx + ib + 6 * (size_t)j;
			pi[0] = (uint16_t)id0; pi[1] = (uint16_t)id2; pi[2] = (uint16_t)id1;
			pi[3] = (uint16_t)id2; pi[4] = (uint16_t)id3; pi[5] = (uint16_t)id1;
		}
		vb += 2 * n; ib += 6 * n;
	}
	memcpy(pos + 2 * (size_t)vb, tessPos, (size_t)numTessVerts * 2 * sizeof(float)); // stroker.cpp:976-994
	for (uint32_t i = 0; i < numTessVerts; ++i) { col[vb + i] = color; }
	const uint16_t delta = (uint16_t)vb; // batchTransformDrawIndices(src, n, dst, (uint16_t)nextVertexID)
	for (uint32_t i = 0; i < numTessIdx; ++i) { idx[ib + i] = (uint16_t)(tessIdx[i] + delta); }
}

} // namespace vgxh
