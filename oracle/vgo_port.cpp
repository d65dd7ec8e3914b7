// vgo_port.cpp -- CPU restatement of the reference's Path flattener and Stroker. See vgo_port.h.
// TEST INFRASTRUCTURE: the "port" oracle. Sequential, scalar, IEEE binary32, no FMA
// (build with -ffp-contract=off). Structure is deliberately different from the reference (one
// run-time parameterised "rails" stroker instead of 48 template instances); the arithmetic and the
// order of every emitted vertex / index follow the cited reference lines exactly.
#include "vgo_port.h"
#include "vgmath.h"
#include <vector>
#include <string.h>

namespace vgo
{
struct P2 { float x, y; };

static inline P2 add(P2 a, P2 b) { return { a.x + b.x, a.y + b.y }; }
static inline P2 sub(P2 a, P2 b) { return { a.x - b.x, a.y - b.y }; }
static inline P2 mul(P2 a, float s) { return { a.x * s, a.y * s }; }
static inline P2 perpCCW(P2 a) { return { -a.y, a.x }; }
static inline P2 perpCW(P2 a) { return { a.y, -a.x }; }
static inline float dot(P2 a, P2 b) { return a.x * b.x + a.y * b.y; }
static inline float cross(P2 a, P2 b) { return a.x * b.y - b.x * a.y; }

// ------------------------------------------------------------------------------------------------
// Path
// ------------------------------------------------------------------------------------------------
struct Path
{
	std::vector<float> verts;  // xy interleaved
	std::vector<SubPath> subs;
	int cur;                   // index of current sub-path or -1 (path.cpp:9 m_CurSubPath)
	uint32_t numVerts;         // path.cpp:10 m_NumVertices (can be < verts.size()/2 after pathClose pops)
	float scale, tol;
};

Path* createPath(bx::AllocatorI*)
{
	Path* p = new Path();
	p->cur = -1;
	p->numVerts = 0;
	p->scale = 1.0f;  // path.cpp:28
	p->tol = 0.25f;   // path.cpp:29
	return p;
}

void destroyPath(Path* path) { delete path; }

void pathReset(Path* path, float scale, float tol) // path.cpp:44-60
{
	path->scale = scale;
	path->tol = tol;
	path->subs.clear();
	path->numVerts = 0;
	path->cur = -1;
}

// path.cpp:748-759: reserve n raw vertices at the end of the vertex array (no dedup).
static float* rawAppend(Path* path, uint32_t n)
{
	path->verts.resize((size_t)(path->numVerts + n) * 2);
	float* p = &path->verts[(size_t)path->numVerts * 2];
	path->numVerts += n;
	return p;
}

static inline bool hasOpenVertices(const Path* path) { return path->cur >= 0 && path->subs[path->cur].m_NumVertices != 0; }

static inline P2 lastVertex(const Path* path)
{
	const SubPath& sp = path->subs[path->cur];
	const uint32_t id = sp.m_FirstVertexID + sp.m_NumVertices - 1;
	return { path->verts[(size_t)id * 2], path->verts[(size_t)id * 2 + 1] };
}

// path.cpp:761-784: epsilon-dedup against the last vertex of the current sub-path.
static void addVertex(Path* path, float x, float y)
{
	SubPath& sp = path->subs[path->cur];
	if (sp.m_NumVertices != 0) {
		const P2 last = lastVertex(path);
		const float dx = last.x - x;
		const float dy = last.y - y;
		if (dx * dx + dy * dy < VGM_EPSILON) {
			return;
		}
	}
	float* v = rawAppend(path, 1);
	v[0] = x;
	v[1] = y;
	path->subs[path->cur].m_NumVertices++;
}

void pathMoveTo(Path* path, float x, float y) // path.cpp:62-78
{
	if (path->cur < 0 || path->subs[path->cur].m_NumVertices != 0) {
		SubPath sp;
		sp.m_FirstVertexID = path->numVerts;
		sp.m_NumVertices = 0;
		sp.m_IsClosed = false;
		path->subs.push_back(sp);
		path->cur = (int)path->subs.size() - 1;
	}
	addVertex(path, x, y);
}

void pathLineTo(Path* path, float x, float y) { addVertex(path, x, y); } // path.cpp:80-84

// path.cpp:86-182. Adaptive de Casteljau with a 10-entry stack of pending right halves; when the
// stack is full the current piece is silently dropped (path.cpp:168-179).
void pathCubicTo(Path* path, float c1x, float c1y, float c2x, float c2y, float x, float y)
{
	struct Piece { float x1, y1, x2, y2, x3, y3, x4, y4; };
	const int kMaxPending = 10; // path.cpp:90
	Piece pending[kMaxPending];
	int numPending = 0;

	const P2 start = lastVertex(path);
	Piece c = { start.x, start.y, c1x, c1y, c2x, c2y, x, y };
	const float tessTol = path->tol / (path->scale * path->scale); // path.cpp:105

	for (;;) {
		const float dx = c.x4 - c.x1;
		const float dy = c.y4 - c.y1;
		const float d2 = vgm_abs((c.x2 - c.x4) * dy - (c.y2 - c.y4) * dx);
		const float d3 = vgm_abs((c.x3 - c.x4) * dy - (c.y3 - c.y4) * dx);
		const float d23 = d2 + d3;
		const bool flat = d23 * d23 <= tessTol * (dx * dx + dy * dy); // path.cpp:116

		if (flat) {
			addVertex(path, c.x4, c.y4);
		} else if (numPending < kMaxPending) {
			const float x12 = (c.x1 + c.x2) * 0.5f, y12 = (c.y1 + c.y2) * 0.5f; // path.cpp:136-147
			const float x23 = (c.x2 + c.x3) * 0.5f, y23 = (c.y2 + c.y3) * 0.5f;
			const float x34 = (c.x3 + c.x4) * 0.5f, y34 = (c.y3 + c.y4) * 0.5f;
			const float x123 = (x12 + x23) * 0.5f, y123 = (y12 + y23) * 0.5f;
			const float x234 = (x23 + x34) * 0.5f, y234 = (y23 + y34) * 0.5f;
			const float x1234 = (x123 + x234) * 0.5f, y1234 = (y123 + y234) * 0.5f;
			pending[numPending++] = { x1234, y1234, x234, y234, x34, y34, c.x4, c.y4 };
			c = { c.x1, c.y1, x12, y12, x123, y123, x1234, y1234 };
			continue;
		}
		// leaf emitted, or piece dropped because the stack is full: continue with the sibling.
		if (numPending == 0) {
			break;
		}
		c = pending[--numPending];
	}
}

void pathQuadraticTo(Path* path, float cx, float cy, float x, float y) // path.cpp:184-201
{
	const P2 p0 = lastVertex(path);
	const float c1x = p0.x + (2.0f / 3.0f) * (cx - p0.x);
	const float c1y = p0.y + (2.0f / 3.0f) * (cy - p0.y);
	const float c2x = x + (2.0f / 3.0f) * (cx - x);
	const float c2y = y + (2.0f / 3.0f) * (cy - y);
	pathCubicTo(path, c1x, c1y, c2x, c2y, x, y);
}

static inline float arcStepAngle(const Path* path, float r) // path.cpp:307, 602, 654
{
	return vgm_acos((path->scale * r) / ((path->scale * r) + path->tol)) * 2.0f;
}

// Rotation recurrence shared by every arc/circle writer (path.cpp:322-336, 609-628, 669-681):
// (ca,sa) <- R(dtheta)(ca,sa), vertex = (cx + rx*ca, cy + ry*sa); written raw (no dedup).
static void appendRotated(Path* path, float cx, float cy, float rx, float ry, float ca, float sa, float cosD, float sinD, uint32_t count)
{
	float* v = rawAppend(path, count);
	for (uint32_t i = 0; i < count; ++i) {
		const float ns = sinD * ca + cosD * sa;
		const float nc = cosD * ca - sinD * sa;
		ca = nc;
		sa = ns;
		v[0] = cx + rx * ca;
		v[1] = cy + ry * sa;
		v += 2;
	}
	path->subs[path->cur].m_NumVertices += count;
}

void pathArc(Path* path, float cx, float cy, float r, float a0, float a1, Winding::Enum dir) // path.cpp:633-682
{
	while (a0 > VGM_PI2) { a0 -= VGM_PI2; }
	while (a1 > VGM_PI2) { a1 -= VGM_PI2; }
	if (dir == Winding::CCW) {
		while (a0 < a1) { a0 += VGM_PI2; }
	} else {
		while (a1 < a0) { a1 += VGM_PI2; }
	}
	const float da = arcStepAngle(path, r);
	const uint32_t numPoints = vgm_umax(2, (uint32_t)vgm_ceil(vgm_abs(a1 - a0) / da));
	const float dtheta = (a1 - a0) / (float)numPoints;
	const float cosD = vgm_cos(dtheta);
	const float sinD = vgm_sin(dtheta);
	const float ca = vgm_cos(a0);
	const float sa = vgm_sin(a0);
	if (hasOpenVertices(path)) {
		pathLineTo(path, cx + r * ca, cy + r * sa);
	} else {
		pathMoveTo(path, cx + r * ca, cy + r * sa);
	}
	appendRotated(path, cx, cy, r, r, ca, sa, cosD, sinD, numPoints);
}

void pathArcTo(Path* path, float x1, float y1, float x2, float y2, float r) // path.cpp:203-273
{
	const P2 p0 = lastVertex(path);
	float dx0 = p0.x - x1, dy0 = p0.y - y1;
	float dx1 = x2 - x1, dy1 = y2 - y1;
	{
		const float lenSqr = dx0 * dx0 + dy0 * dy0;
		const float invLen = lenSqr < VGM_EPSILON ? 0.0f : vgm_rsqrt(lenSqr);
		dx0 *= invLen;
		dy0 *= invLen;
	}
	{
		const float lenSqr = dx1 * dx1 + dy1 * dy1;
		const float invLen = lenSqr < VGM_EPSILON ? 0.0f : vgm_rsqrt(lenSqr);
		dx1 *= invLen;
		dy1 *= invLen;
	}
	const float a = vgm_acos(dx0 * dx1 + dy0 * dy1);
	const float d = r / vgm_tan(a / 2.0f);
	if (d > 10000.0f) {
		pathLineTo(path, x1, y1);
		return;
	}
	float cx, cy, a0, a1;
	Winding::Enum dir;
	const float crs = dx1 * dy0 - dx0 * dy1;
	if (crs > 0.0f) {
		cx = x1 + dx0 * d + dy0 * r;
		cy = y1 + dy0 * d - dx0 * r;
		a0 = vgm_atan2(dx0, -dy0);
		a1 = vgm_atan2(-dx1, dy1);
		dir = Winding::CW;
	} else {
		cx = x1 + dx0 * d - dy0 * r;
		cy = y1 + dy0 * d + dx0 * r;
		a0 = vgm_atan2(-dx0, dy0);
		a1 = vgm_atan2(dx1, -dy1);
		dir = Winding::CCW;
	}
	pathArc(path, cx, cy, r, a0, a1, dir);
}

void pathRect(Path* path, float x, float y, float w, float h) // path.cpp:275-286
{
	if (vgm_abs(w) < VGM_EPSILON || vgm_abs(h) < VGM_EPSILON) {
		return;
	}
	pathMoveTo(path, x, y);
	pathLineTo(path, x, y + h);
	pathLineTo(path, x + w, y + h);
	pathLineTo(path, x + w, y);
	pathClose(path);
}

void pathRoundedRect(Path* path, float x, float y, float w, float h, float r) // path.cpp:288-409
{
	if (r < 0.1f) {
		pathRect(path, x, y, w, h);
		return;
	}
	const float maxR = vgm_min(w, h) * 0.5f;
	if (w == h && r >= maxR - VGM_EPSILON) {
		pathCircle(path, x + maxR, y + maxR, maxR);
		return;
	}
	r = vgm_min(r, maxR);
	const float da = arcStepAngle(path, r);
	const uint32_t half = vgm_umax(2, (uint32_t)vgm_ceil(VGM_PI / da));
	const uint32_t quarter = (half >> 1) + 1;
	const float dtheta = -VGM_PIHALF / (float)(quarter - 1);
	const float cosD = vgm_cos(dtheta);
	const float sinD = vgm_sin(dtheta);

	pathMoveTo(path, x, y + r);
	pathLineTo(path, x, y + h - r);
	appendRotated(path, x + r, y + h - r, r, r, -1.0f, 0.0f, cosD, sinD, quarter - 1);     // bottom left
	pathLineTo(path, x + w - r, y + h);
	appendRotated(path, x + w - r, y + h - r, r, r, 0.0f, 1.0f, cosD, sinD, quarter - 1);  // bottom right
	pathLineTo(path, x + w, y + r);
	appendRotated(path, x + w - r, y + r, r, r, 1.0f, 0.0f, cosD, sinD, quarter - 1);      // top right
	pathLineTo(path, x + r, y);
	appendRotated(path, x + r, y + r, r, r, 0.0f, -1.0f, cosD, sinD, quarter - 1);         // top left
	pathClose(path);
}

// One corner of pathRoundedRectVarying (path.cpp:428-455 and its three siblings).
static void variedCorner(Path* path, float rc, float cx, float cy, float ca, float sa)
{
	const float halfDa = vgm_acos((path->scale * rc) / ((path->scale * rc) + path->tol));
	const uint32_t half = vgm_umax(2, (uint32_t)vgm_ceil(VGM_PIHALF / halfDa));
	const uint32_t quarter = (half >> 1) + 1;
	const float dtheta = -VGM_PIHALF / (float)(quarter - 1);
	appendRotated(path, cx, cy, rc, rc, ca, sa, vgm_cos(dtheta), vgm_sin(dtheta), quarter - 1);
}

void pathRoundedRectVarying(Path* path, float x, float y, float w, float h, float rTL, float rTR, float rBR, float rBL) // path.cpp:411-559
{
	if (rTL < 0.1f && rBL < 0.1f && rBR < 0.1f && rTR < 0.1f) {
		pathRect(path, x, y, w, h);
		return;
	}
	const float halfw = w * 0.5f;
	const float halfh = h * 0.5f;
	const float rtl = vgm_min(vgm_min(rTL, halfw), halfh);
	const float rtr = vgm_min(vgm_min(rTR, halfw), halfh);
	const float rbl = vgm_min(vgm_min(rBL, halfw), halfh);
	const float rbr = vgm_min(vgm_min(rBR, halfw), halfh);

	if (rtl < 0.1f) {
		pathMoveTo(path, x, y);
	} else {
		pathMoveTo(path, x + rtl, y);
		variedCorner(path, rtl, x + rtl, y + rtl, 0.0f, -1.0f);
	}
	if (rbl < 0.1f) {
		pathLineTo(path, x, y + h);
	} else {
		pathLineTo(path, x, y + h - rbl);
		variedCorner(path, rbl, x + rbl, y + h - rbl, -1.0f, 0.0f);
	}
	if (rbr < 0.1f) {
		pathLineTo(path, x + w, y + h);
	} else {
		pathLineTo(path, x + w - rbr, y + h);
		variedCorner(path, rbr, x + w - rbr, y + h - rbr, 0.0f, 1.0f);
	}
	if (rtr < 0.1f) {
		pathLineTo(path, x + w, y);
	} else {
		pathLineTo(path, x + w, y + rtr);
		variedCorner(path, rtr, x + w - rtr, y + rtr, 1.0f, 0.0f);
	}
	pathClose(path);
}

void pathCircle(Path* path, float cx, float cy, float r) { pathEllipse(path, cx, cy, r, r); } // path.cpp:561-564

void pathEllipse(Path* path, float cx, float cy, float rx, float ry) // path.cpp:599-631
{
	const float avgR = (rx + ry) * 0.5f;
	const float da = arcStepAngle(path, avgR);
	const uint32_t half = vgm_umax(2, (uint32_t)vgm_ceil(VGM_PI / da));
	const uint32_t numPoints = half * 2;
	pathMoveTo(path, cx + rx, cy);
	const float dtheta = -VGM_PI2 / (float)numPoints;
	appendRotated(path, cx, cy, rx, ry, 1.0f, 0.0f, vgm_cos(dtheta), vgm_sin(dtheta), numPoints - 1);
	pathClose(path);
}

void pathPolyline(Path* path, const float* coords, uint32_t numPoints) // path.cpp:684-705
{
	if (path->subs[path->cur].m_NumVertices > 0) {
		const P2 last = lastVertex(path);
		const float dx = last.x - coords[0];
		const float dy = last.y - coords[1];
		if (dx * dx + dy * dy < VGM_EPSILON) {
			coords += 2;
			numPoints--;
		}
	}
	float* v = rawAppend(path, numPoints);
	memcpy(v, coords, sizeof(float) * 2 * numPoints);
	path->subs[path->cur].m_NumVertices += numPoints;
}

void pathClose(Path* path) // path.cpp:707-726
{
	SubPath& sp = path->subs[path->cur];
	if (sp.m_IsClosed || sp.m_NumVertices <= 2) {
		return;
	}
	sp.m_IsClosed = true;
	const float* first = &path->verts[(size_t)sp.m_FirstVertexID * 2];
	const float* last = &path->verts[(size_t)(sp.m_FirstVertexID + sp.m_NumVertices - 1) * 2];
	const float dx = last[0] - first[0];
	const float dy = last[1] - first[1];
	if (dx * dx + dy * dy < VGM_EPSILON) {
		--sp.m_NumVertices;
		--path->numVerts;
	}
}

const float* pathGetVertices(const Path* path) { return path->verts.data(); }
uint32_t pathGetNumVertices(const Path* path) { return path->numVerts; }
const SubPath* pathGetSubPaths(const Path* path) { return path->subs.data(); }
uint32_t pathGetNumSubPaths(const Path* path) { return (uint32_t)path->subs.size(); }

void batchTransformPositions(const float* v, uint32_t n, float* p, const float* m)
{
	for (uint32_t i = 0; i < n; ++i) {
		const float x = v[i * 2], y = v[i * 2 + 1];
		p[i * 2] = m[0] * x + m[2] * y + m[4];     // vg_util.h:26
		p[i * 2 + 1] = m[1] * x + m[3] * y + m[5]; // vg_util.h:27
	}
}

// ------------------------------------------------------------------------------------------------
// Stroker
// ------------------------------------------------------------------------------------------------
struct Stroker
{
	std::vector<P2> pos;
	std::vector<uint32_t> col;
	std::vector<uint16_t> idx;
	float fringe, scale, tol;

	void reset() { pos.clear(); col.clear(); idx.clear(); }                 // stroker.cpp:2316-2320
	uint16_t nextID() const { return (uint16_t)pos.size(); }                // (uint16_t)m_NumVertices casts
	void v(P2 p) { pos.push_back(p); }
	void vc(P2 p, uint32_t c) { pos.push_back(p); col.push_back(c); }
	void tri(uint32_t a, uint32_t b, uint32_t c) { idx.push_back((uint16_t)a); idx.push_back((uint16_t)b); idx.push_back((uint16_t)c); }
	void finish(Mesh* m, bool withColor)
	{
		m->m_PosBuffer = pos.empty() ? nullptr : &pos[0].x;
		m->m_ColorBuffer = withColor ? col.data() : nullptr;
		m->m_IndexBuffer = idx.data();
		m->m_NumVertices = (uint32_t)pos.size();
		m->m_NumIndices = (uint32_t)idx.size();
	}
};

Stroker* createStroker(bx::AllocatorI*)
{
	Stroker* s = new Stroker();
	s->fringe = 1.0f; // stroker.cpp:199-201
	s->scale = 1.0f;
	s->tol = 0.25f;
	return s;
}

void destroyStroker(Stroker* s) { delete s; }

void strokerReset(Stroker* s, float scale, float tol, float fringe) // stroker.cpp:232-237
{
	s->scale = scale;
	s->tol = tol;
	s->fringe = fringe;
}

static inline P2 dirTo(P2 a, P2 b) // stroker.cpp:31-38
{
	const float dx = b.x - a.x;
	const float dy = b.y - a.y;
	const float lenSqr = dx * dx + dy * dy;
	const float invLen = lenSqr < VGM_EPSILON ? 0.0f : vgm_rsqrt(lenSqr);
	return { dx * invLen, dy * invLen };
}

static inline P2 extrusion(P2 d01, P2 d12) // stroker.cpp:40-53
{
	P2 v = perpCCW(d01);
	const float c = cross(d12, d01);
	if (vgm_abs(c) > (1.0f / 100.0f)) {
		v = mul(sub(d01, d12), 1.0f / c);
	}
	return v;
}

static inline uint32_t halfCirclePoints(const Stroker* s, float hsw) // stroker.cpp:1013-1014, 1398-1399
{
	const float da = vgm_acos((s->scale * hsw) / ((s->scale * hsw) + s->tol)) * 2.0f;
	return vgm_umax(2u, (uint32_t)vgm_ceil(VGM_PI / da));
}

static inline float stepAngle(const Stroker* s, float hsw)
{
	return vgm_acos((s->scale * hsw) / ((s->scale * hsw) + s->tol)) * 2.0f;
}

// Arc description at a Round join (stroker.cpp:1140-1147 / 1238-1245, 1588-1595 / 1744-1751).
struct JoinArc { float a01, arcDa; uint32_t n; };

static inline JoinArc roundJoinArc(P2 n01, P2 n12, bool leftInner, float da)
{
	JoinArc r;
	float a01 = vgm_atan2(n01.y, n01.x);
	float a12 = vgm_atan2(n12.y, n12.x);
	if (leftInner) {
		if (a12 < a01) { a12 += VGM_PI2; }
		r.n = vgm_umax(2u, (uint32_t)((a12 - a01) / da));
	} else {
		if (a12 > a01) { a12 -= VGM_PI2; }
		r.n = vgm_umax(2u, (uint32_t)((a01 - a12) / da));
	}
	r.a01 = a01;
	r.arcDa = (a12 - a01) / (float)r.n;
	return r;
}

static inline bool validCapJoin(uint32_t cap, uint32_t join) { return cap <= 2 && join <= 2; }

// ---- non-AA stroke: 2 rails (stroker.cpp:1008-1388) ---------------------------------------------
struct Rails2 { uint16_t l, r; };
static inline void bridge2(Stroker* s, Rails2 p, Rails2 c) { s->tri(p.l, p.r, c.r); s->tri(p.l, c.r, c.l); }

void strokerPolylineStroke(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, bool closed, float strokeWidth, LineCap::Enum cap, LineJoin::Enum join)
{
	if (!validCapJoin(cap, join)) {
		return; // stroker.cpp:269-271: mesh left untouched
	}
	const P2* vtx = (const P2*)vertexList;
	const uint32_t numSegments = n - (closed ? 0 : 1);
	const float hsw = strokeWidth * 0.5f;
	const float da = stepAngle(s, hsw);
	const uint32_t H = halfCirclePoints(s, hsw);
	s->reset();

	P2 d01;
	Rails2 prev = { 0xFFFF, 0xFFFF }, first = { 0xFFFF, 0xFFFF };
	bool havePrev = false;
	if (!closed) { // first cap, stroker.cpp:1023-1083
		const P2 p0 = vtx[0];
		d01 = dirTo(p0, vtx[1]);
		const P2 l01 = perpCCW(d01);
		if (cap == LineCap::Butt) {
			const P2 lh = mul(l01, hsw);
			s->v(add(p0, lh));
			s->v(sub(p0, lh));
			prev = { 0, 1 };
		} else if (cap == LineCap::Square) {
			const P2 lh = mul(l01, hsw);
			const P2 dh = mul(d01, hsw);
			s->v(add(p0, sub(lh, dh)));
			s->v(sub(p0, add(lh, dh)));
			prev = { 0, 1 };
		} else {
			const float startAngle = vgm_atan2(l01.y, l01.x);
			for (uint32_t i = 0; i < H; ++i) {
				const float a = startAngle + i * VGM_PI / (float)(H - 1);
				const float ca = vgm_cos(a), sa = vgm_sin(a);
				s->v({ p0.x + ca * hsw, p0.y + sa * hsw });
			}
			for (uint32_t i = 0; i < H - 2; ++i) {
				s->tri(0, i + 1, i + 2);
			}
			prev = { 0, (uint16_t)(H - 1) };
		}
		havePrev = true;
	} else {
		d01 = dirTo(vtx[n - 1], vtx[0]);
	}

	for (uint32_t i = closed ? 0 : 1; i < numSegments; ++i) { // stroker.cpp:1088-1296
		const P2 p1 = vtx[i];
		const P2 p2 = vtx[i == n - 1 ? 0 : i + 1];
		const P2 d12 = dirTo(p1, p2);
		const P2 vh = mul(extrusion(d01, d12), hsw);
		const bool leftInner = (d12.x * vh.x + d12.y * vh.y) >= 0.0f;
		const uint16_t b = s->nextID();
		Rails2 entry, exit;
		const P2 inner = leftInner ? add(p1, vh) : sub(p1, vh);
		const P2 outer = leftInner ? sub(p1, vh) : add(p1, vh);
		if (join == LineJoin::Miter) {
			s->v(inner);
			s->v(outer);
			entry = leftInner ? Rails2{ b, (uint16_t)(b + 1) } : Rails2{ (uint16_t)(b + 1), b };
			exit = entry;
			if (havePrev) { bridge2(s, prev, entry); } else { first = entry; }
		} else {
			const P2 n01 = leftInner ? perpCW(d01) : perpCCW(d01);
			const P2 n12 = leftInner ? perpCW(d12) : perpCCW(d12);
			JoinArc arc = { 0.0f, 0.0f, 1 };
			if (join == LineJoin::Round) {
				arc = roundJoinArc(n01, n12, leftInner, da);
			}
			s->v(inner);
			s->v(add(p1, mul(n01, hsw)));
			for (uint32_t k = 1; k < arc.n; ++k) {
				const float a = arc.a01 + k * arc.arcDa;
				const float ca = vgm_cos(a), sa = vgm_sin(a);
				s->v({ p1.x + hsw * ca, p1.y + hsw * sa });
			}
			s->v(add(p1, mul(n12, hsw)));
			entry = leftInner ? Rails2{ b, (uint16_t)(b + 1) } : Rails2{ (uint16_t)(b + 1), b };
			if (havePrev) { bridge2(s, prev, entry); } else { first = entry; }
			for (uint32_t k = 0; k < arc.n; ++k) {
				const uint16_t base = b + (uint16_t)k;
				if (leftInner) { s->tri(b, (uint16_t)(base + 1), (uint16_t)(base + 2)); }
				else { s->tri(b, (uint16_t)(base + 2), (uint16_t)(base + 1)); }
			}
			const uint16_t endID = b + (uint16_t)arc.n + 1;
			exit = leftInner ? Rails2{ b, endID } : Rails2{ endID, b };
		}
		prev = exit;
		havePrev = true;
		d01 = d12;
	}

	if (!closed) { // last cap, stroker.cpp:1298-1371
		const P2 p1 = vtx[n - 1];
		const P2 l01 = perpCCW(d01);
		const uint16_t c = s->nextID();
		if (cap == LineCap::Butt) {
			const P2 lh = mul(l01, hsw);
			s->v(add(p1, lh));
			s->v(sub(p1, lh));
			bridge2(s, prev, { c, (uint16_t)(c + 1) });
		} else if (cap == LineCap::Square) {
			const P2 lh = mul(l01, hsw);
			const P2 dh = mul(d01, hsw);
			s->v(add(p1, add(lh, dh)));
			s->v(sub(p1, sub(lh, dh)));
			bridge2(s, prev, { c, (uint16_t)(c + 1) });
		} else {
			const float startAngle = vgm_atan2(l01.y, l01.x);
			for (uint32_t i = 0; i < H; ++i) {
				const float a = startAngle - i * VGM_PI / (float)(H - 1);
				const float ca = vgm_cos(a), sa = vgm_sin(a);
				s->v({ p1.x + ca * hsw, p1.y + sa * hsw });
			}
			bridge2(s, prev, { c, (uint16_t)(c + (H - 1)) });
			for (uint32_t i = 0; i < H - 2; ++i) {
				const uint16_t base = c + (uint16_t)i;
				s->tri(c, (uint16_t)(base + 2), (uint16_t)(base + 1));
			}
		}
	} else {
		bridge2(s, prev, first); // stroker.cpp:1372-1380
	}
	s->finish(mesh, false);
}

// ---- AA stroke: 4 rails (stroker.cpp:1390-1991) -------------------------------------------------
struct Rails4 { uint16_t laa, l, r, raa; };
static inline void bridge4(Stroker* s, Rails4 p, Rails4 c)
{
	s->tri(p.laa, p.l, c.l);
	s->tri(p.laa, c.l, c.laa);
	s->tri(p.l, p.r, c.r);
	s->tri(p.l, c.r, c.l);
	s->tri(p.r, p.raa, c.raa);
	s->tri(p.r, c.raa, c.r);
}

void strokerPolylineStrokeAA(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, bool closed, Color color, float strokeWidth, LineCap::Enum cap, LineJoin::Enum join)
{
	if (!validCapJoin(cap, join)) {
		return; // stroker.cpp:305-307
	}
	const P2* vtx = (const P2*)vertexList;
	const uint32_t numSegments = n - (closed ? 0 : 1);
	const uint32_t c0 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
	const float fringe = s->fringe;
	const float hsw = (strokeWidth - fringe) * 0.5f;
	const float hswAA = hsw + fringe;
	const float da = stepAngle(s, hsw);
	const uint32_t H = halfCirclePoints(s, hsw);
	s->reset();

	P2 d01;
	Rails4 prev = { 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF }, first = prev;
	bool havePrev = false;
	if (!closed) { // first cap, stroker.cpp:1413-1518
		const P2 p0 = vtx[0];
		d01 = dirTo(p0, vtx[1]);
		const P2 l01 = perpCCW(d01);
		if (cap == LineCap::Butt) {
			const P2 lh = mul(l01, hsw);
			const P2 lhaa = mul(l01, hswAA);
			const P2 daa = mul(d01, fringe);
			s->vc(add(p0, sub(lhaa, daa)), c0);
			s->vc(add(p0, lh), color);
			s->vc(sub(p0, lh), color);
			s->vc(sub(p0, add(lhaa, daa)), c0);
			s->tri(0, 2, 1);
			s->tri(0, 3, 2);
			prev = { 0, 1, 2, 3 };
		} else if (cap == LineCap::Square) {
			const P2 lh = mul(l01, hsw);
			const P2 dh = mul(d01, hsw);
			const P2 lhaa = mul(l01, hswAA);
			const P2 dhaa = mul(d01, hswAA);
			s->vc(add(p0, sub(lhaa, dhaa)), c0);
			s->vc(add(p0, sub(lh, dh)), color);
			s->vc(sub(p0, add(lh, dh)), color);
			s->vc(sub(p0, add(lhaa, dhaa)), c0);
			s->tri(0, 2, 1);
			s->tri(0, 3, 2);
			prev = { 0, 1, 2, 3 };
		} else {
			const float startAngle = vgm_atan2(l01.y, l01.x);
			for (uint32_t i = 0; i < H; ++i) {
				const float a = startAngle + i * VGM_PI / (float)(H - 1);
				const float ca = vgm_cos(a), sa = vgm_sin(a);
				s->vc({ p0.x + ca * hsw, p0.y + sa * hsw }, color);
				s->vc({ p0.x + ca * hswAA, p0.y + sa * hswAA }, c0);
			}
			for (uint32_t i = 0; i < H - 2; ++i) {
				s->tri(0, (i << 1) + 2, (i << 1) + 4);
			}
			for (uint32_t i = 0; i < H - 1; ++i) {
				const uint16_t base = (uint16_t)(i << 1);
				s->tri(base, (uint16_t)(base + 1), (uint16_t)(base + 3));
				s->tri(base, (uint16_t)(base + 3), (uint16_t)(base + 2));
			}
			prev = { 1, 0, (uint16_t)((H - 1) * 2), (uint16_t)((H - 1) * 2 + 1) };
		}
		havePrev = true;
	} else {
		d01 = dirTo(vtx[n - 1], vtx[0]);
	}

	for (uint32_t i = closed ? 0 : 1; i < numSegments; ++i) { // stroker.cpp:1520-1850
		const P2 p1 = vtx[i];
		const P2 p2 = vtx[i == n - 1 ? 0 : i + 1];
		const P2 d12 = dirTo(p1, p2);
		const P2 v = extrusion(d01, d12);
		const P2 vhaa = mul(v, hswAA);
		const bool leftInner = (d12.x * vhaa.x + d12.y * vhaa.y) >= 0.0f;
		const P2 vh = mul(v, hsw);
		const uint16_t b = s->nextID();
		const P2 innerAA = leftInner ? add(p1, vhaa) : sub(p1, vhaa);
		const P2 inner = leftInner ? add(p1, vh) : sub(p1, vh);
		const Rails4 entry = leftInner ? Rails4{ b, (uint16_t)(b + 1), (uint16_t)(b + 2), (uint16_t)(b + 3) }
		                               : Rails4{ (uint16_t)(b + 3), (uint16_t)(b + 2), (uint16_t)(b + 1), b };
		Rails4 exit = entry;
		if (join == LineJoin::Miter) {
			s->vc(innerAA, c0);
			s->vc(inner, color);
			s->vc(leftInner ? sub(p1, vh) : add(p1, vh), color);
			s->vc(leftInner ? sub(p1, vhaa) : add(p1, vhaa), c0);
			if (havePrev) { bridge4(s, prev, entry); } else { first = entry; }
		} else {
			const P2 n01 = leftInner ? perpCW(d01) : perpCCW(d01);
			const P2 n12 = leftInner ? perpCW(d12) : perpCCW(d12);
			JoinArc arc = { 0.0f, 0.0f, 1 };
			if (join == LineJoin::Round) {
				arc = roundJoinArc(n01, n12, leftInner, da);
			}
			s->vc(innerAA, c0);
			s->vc(inner, color);
			{ // first arc vertex pair
				P2 a = add(p1, mul(n01, hsw));
				const P2 aAA = add(p1, mul(n01, hswAA));
				if (join == LineJoin::Bevel) {
					const float cosAngle = vgm_abs(dot(n01, n12));
					a = sub(a, mul(d01, cosAngle * fringe));
				}
				s->vc(a, color);
				s->vc(aAA, c0);
			}
			for (uint32_t k = 1; k < arc.n; ++k) {
				const float a = arc.a01 + k * arc.arcDa;
				const P2 dir = { vgm_cos(a), vgm_sin(a) };
				s->vc(add(p1, mul(dir, hsw)), color);
				s->vc(add(p1, mul(dir, hswAA)), c0);
			}
			{ // last arc vertex pair
				P2 a = add(p1, mul(n12, hsw));
				const P2 aAA = add(p1, mul(n12, hswAA));
				if (join == LineJoin::Bevel) {
					const float cosAngle = vgm_abs(dot(n01, n12));
					a = add(a, mul(d12, cosAngle * fringe));
				}
				s->vc(a, color);
				s->vc(aAA, c0);
			}
			if (havePrev) { bridge4(s, prev, entry); } else { first = entry; }
			uint16_t arcID = b + 2;
			for (uint32_t k = 0; k < arc.n; ++k) {
				if (leftInner) {
					s->tri((uint16_t)(b + 1), arcID, (uint16_t)(arcID + 2));
					s->tri(arcID, (uint16_t)(arcID + 1), (uint16_t)(arcID + 3));
					s->tri(arcID, (uint16_t)(arcID + 3), (uint16_t)(arcID + 2));
				} else {
					s->tri((uint16_t)(b + 1), (uint16_t)(arcID + 2), arcID);
					s->tri(arcID, (uint16_t)(arcID + 3), (uint16_t)(arcID + 1));
					s->tri(arcID, (uint16_t)(arcID + 2), (uint16_t)(arcID + 3));
				}
				arcID += 2;
			}
			exit = leftInner ? Rails4{ b, (uint16_t)(b + 1), arcID, (uint16_t)(arcID + 1) }
			                 : Rails4{ (uint16_t)(arcID + 1), arcID, (uint16_t)(b + 1), b };
		}
		prev = exit;
		havePrev = true;
		d01 = d12;
	}

	if (!closed) { // last cap, stroker.cpp:1852-1969
		const P2 p1 = vtx[n - 1];
		const P2 l01 = perpCCW(d01);
		const uint16_t c = s->nextID();
		if (cap == LineCap::Butt || cap == LineCap::Square) {
			if (cap == LineCap::Butt) {
				const P2 lh = mul(l01, hsw);
				const P2 lhaa = mul(l01, hswAA);
				const P2 daa = mul(d01, fringe);
				s->vc(add(p1, add(lhaa, daa)), c0);
				s->vc(add(p1, lh), color);
				s->vc(sub(p1, lh), color);
				s->vc(sub(p1, sub(lhaa, daa)), c0);
			} else {
				const P2 lh = mul(l01, hsw);
				const P2 dh = mul(d01, hsw);
				const P2 lhaa = mul(l01, hswAA);
				const P2 dhaa = mul(d01, hswAA);
				s->vc(add(p1, add(lhaa, dhaa)), c0);
				s->vc(add(p1, add(lh, dh)), color);
				s->vc(sub(p1, sub(lh, dh)), color);
				s->vc(sub(p1, sub(lhaa, dhaa)), c0);
			}
			bridge4(s, prev, { c, (uint16_t)(c + 1), (uint16_t)(c + 2), (uint16_t)(c + 3) });
			s->tri(c, (uint16_t)(c + 1), (uint16_t)(c + 2));
			s->tri(c, (uint16_t)(c + 2), (uint16_t)(c + 3));
		} else {
			const float startAngle = vgm_atan2(l01.y, l01.x);
			for (uint32_t i = 0; i < H; ++i) {
				const float a = startAngle - i * VGM_PI / (float)(H - 1);
				const float ca = vgm_cos(a), sa = vgm_sin(a);
				s->vc({ p1.x + ca * hsw, p1.y + sa * hsw }, color);
				s->vc({ p1.x + ca * hswAA, p1.y + sa * hswAA }, c0);
			}
			const uint16_t e = (uint16_t)(c + (H - 1) * 2);
			bridge4(s, prev, { (uint16_t)(c + 1), c, e, (uint16_t)(e + 1) });
			for (uint32_t i = 0; i < H - 2; ++i) {
				const uint16_t base = c + (uint16_t)(i << 1);
				s->tri(c, (uint16_t)(base + 4), (uint16_t)(base + 2));
			}
			for (uint32_t i = 0; i < H - 1; ++i) {
				const uint16_t base = c + (uint16_t)(i << 1);
				s->tri(base, (uint16_t)(base + 3), (uint16_t)(base + 1));
				s->tri(base, (uint16_t)(base + 2), (uint16_t)(base + 3));
			}
		}
	} else {
		bridge4(s, prev, first); // stroker.cpp:1970-1984
	}
	s->finish(mesh, true);
}

// ---- thin AA stroke: 3 rails (stroker.cpp:1993-2314) --------------------------------------------
struct Rails3 { uint16_t laa, m, raa; };
static inline void bridge3(Stroker* s, Rails3 p, Rails3 c)
{
	s->tri(p.laa, p.m, c.m);
	s->tri(p.laa, c.m, c.laa);
	s->tri(p.m, p.raa, c.raa);
	s->tri(p.m, c.raa, c.m);
}

void strokerPolylineStrokeAAThin(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, bool closed, Color color, LineCap::Enum capIn, LineJoin::Enum joinIn)
{
	// dispatch table stroker.cpp:311-332: perm = cap | join << 2; Round cap -> Square, Round join -> Bevel.
	if (!validCapJoin(capIn, joinIn)) {
		return;
	}
	const bool squareCap = capIn != LineCap::Butt;
	const bool bevel = joinIn != LineJoin::Miter;
	const P2* vtx = (const P2*)vertexList;
	const uint32_t numSegments = n - (closed ? 0 : 1);
	const uint32_t c0 = color & 0x00FFFFFFu;
	const float f = s->fringe; // stroker.cpp:1999
	s->reset();

	P2 d01;
	Rails3 prev = { 0xFFFF, 0xFFFF, 0xFFFF }, first = prev;
	bool havePrev = false;
	if (!closed) { // stroker.cpp:2012-2058
		const P2 p0 = vtx[0];
		d01 = dirTo(p0, vtx[1]);
		const P2 l01 = perpCCW(d01);
		const P2 lf = mul(l01, f);
		if (!squareCap) {
			s->vc(add(p0, lf), c0);
			s->vc(p0, color);
			s->vc(sub(p0, lf), c0);
		} else {
			const P2 df = mul(d01, f);
			s->vc(add(p0, sub(lf, df)), c0);
			s->vc(p0, color);
			s->vc(sub(p0, add(lf, df)), c0);
		}
		prev = { 0, 1, 2 };
		havePrev = true;
	} else {
		d01 = dirTo(vtx[n - 1], vtx[0]);
	}

	for (uint32_t i = closed ? 0 : 1; i < numSegments; ++i) { // stroker.cpp:2060-2240
		const P2 p1 = vtx[i];
		const P2 p2 = vtx[i == n - 1 ? 0 : i + 1];
		const P2 d12 = dirTo(p1, p2);
		const P2 vf = mul(extrusion(d01, d12), f);
		const bool leftInner = (d12.x * vf.x + d12.y * vf.y) >= 0.0f;
		const uint16_t b = s->nextID();
		const P2 inner = leftInner ? add(p1, vf) : sub(p1, vf);
		const Rails3 entry = leftInner ? Rails3{ b, (uint16_t)(b + 1), (uint16_t)(b + 2) }
		                               : Rails3{ (uint16_t)(b + 2), (uint16_t)(b + 1), b };
		Rails3 exit = entry;
		if (!bevel) {
			s->vc(inner, c0);
			s->vc(p1, color);
			s->vc(leftInner ? sub(p1, vf) : add(p1, vf), c0);
			if (havePrev) { bridge3(s, prev, entry); } else { first = entry; }
		} else {
			const P2 n01 = leftInner ? perpCW(d01) : perpCCW(d01);
			const P2 n12 = leftInner ? perpCW(d12) : perpCCW(d12);
			s->vc(inner, c0);
			s->vc(p1, color);
			s->vc(add(p1, mul(n01, f)), c0);
			s->vc(add(p1, mul(n12, f)), c0);
			if (havePrev) { bridge3(s, prev, entry); } else { first = entry; }
			if (leftInner) {
				s->tri((uint16_t)(b + 1), (uint16_t)(b + 2), (uint16_t)(b + 3));
				exit = { b, (uint16_t)(b + 1), (uint16_t)(b + 3) };
			} else {
				s->tri((uint16_t)(b + 1), (uint16_t)(b + 3), (uint16_t)(b + 2));
				exit = { (uint16_t)(b + 3), (uint16_t)(b + 1), b };
			}
		}
		prev = exit;
		havePrev = true;
		d01 = d12;
	}

	if (!closed) { // stroker.cpp:2242-2294
		const P2 p1 = vtx[n - 1];
		const P2 l01 = perpCCW(d01);
		const uint16_t c = s->nextID();
		const P2 lf = mul(l01, f);
		if (!squareCap) {
			s->vc(add(p1, lf), c0);
			s->vc(p1, color);
			s->vc(sub(p1, lf), c0);
		} else {
			const P2 df = mul(d01, f);
			s->vc(add(p1, add(lf, df)), c0);
			s->vc(p1, color);
			s->vc(sub(p1, sub(lf, df)), c0);
		}
		bridge3(s, prev, { c, (uint16_t)(c + 1), (uint16_t)(c + 2) });
	} else {
		bridge3(s, prev, first); // stroker.cpp:2295-2306
	}
	s->finish(mesh, true);
}

// ---- convex fills -------------------------------------------------------------------------------
void strokerConvexFill(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n) // stroker.cpp:334-365
{
	s->reset();
	for (uint32_t t = 0; t + 2 < n; ++t) {
		s->tri(0, t + 1, t + 2);
	}
	mesh->m_PosBuffer = vertexList; // aliased, stroker.cpp:360
	mesh->m_ColorBuffer = nullptr;
	mesh->m_IndexBuffer = s->idx.data();
	mesh->m_NumVertices = n;
	mesh->m_NumIndices = (uint32_t)s->idx.size();
}

void strokerConvexFillAA(Stroker* s, Mesh* mesh, const float* vertexList, uint32_t n, uint32_t color) // stroker.cpp:713-807
{
	const P2* vtx = (const P2*)vertexList;
	const float orient = cross(sub(vtx[1], vtx[0]), sub(vtx[2], vtx[0])); // first triangle only, :721
	const float aa = s->fringe * 0.5f * vgm_sign(orient);
	const uint32_t c0 = color & 0x00FFFFFFu;
	s->reset();

	P2 d01 = dirTo(vtx[n - 1], vtx[0]);
	for (uint32_t i = 0; i < n; ++i) {
		const P2 p1 = vtx[i];
		const P2 p2 = vtx[i == n - 1 ? 0 : i + 1];
		const P2 d12 = dirTo(p1, p2);
		const P2 vaa = mul(extrusion(d01, d12), aa);
		s->vc(add(p1, vaa), color);
		s->vc(sub(p1, vaa), c0);
		d01 = d12;
	}
	for (uint32_t t = 0; t + 2 < n; ++t) { // fan over the inner (even) vertices, :769-776
		s->tri(0, 2 * t + 2, 2 * t + 4);
	}
	for (uint32_t i = 0; i + 1 < n; ++i) { // fringe quads, :779-787
		const uint32_t b = 2 * i;
		s->tri(b, b + 1, b + 3);
		s->tri(b, b + 3, b + 2);
	}
	const uint32_t b = 2 * (n - 1); // wrap-around quad, :789-795
	s->tri(b, b + 1, 1);
	s->tri(b, 1, 0);
	s->finish(mesh, true);
}
}
