// compat_test.cpp -- drives the SAME sequence of vg::pathXXX / vg::strokerXXX calls through
//   (a) include/vgx_compat.hpp  (product: C++ API over the C-ABI, HIP kernels), and
//   (b) oracle/vgo_port.h       (CPU oracle, test infrastructure)
// and compares every vertex, sub-path, colour and index bit for bit. Built and run by tests/test_gpu_compat.py.
#include "vgx_compat.hpp"
#include "vgo_port.h"
#include <stdio.h>
#include <string.h>
#include <vector>
#include <dlfcn.h>
#include <math.h>

static uint32_t g_rng = 12345u;
static float frand(float lo, float hi) { g_rng = g_rng * 1664525u + 1013904223u; return lo + (hi - lo) * (float)((g_rng >> 8) & 0xFFFFFF) / 16777216.0f; }
static uint32_t irand(uint32_t n) { g_rng = g_rng * 1664525u + 1013904223u; return (g_rng >> 10) % n; }

static int g_fail = 0, g_checks = 0;
#define CHECK(cond, ...) do { ++g_checks; if (!(cond)) { ++g_fail; if (g_fail < 10) { printf("FAIL line %d: ", __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

template<class M1, class M2>
static void compareMesh(const M1& a, const M2& b, bool posAliased, const char* what)
{
	CHECK(a.m_NumVertices == b.m_NumVertices && a.m_NumIndices == b.m_NumIndices, "%s counts %u/%u vs %u/%u", what, a.m_NumVertices, a.m_NumIndices, b.m_NumVertices, b.m_NumIndices);
	if (a.m_NumVertices != b.m_NumVertices || a.m_NumIndices != b.m_NumIndices) { return; }
	CHECK(memcmp(a.m_IndexBuffer, b.m_IndexBuffer, a.m_NumIndices * 2) == 0, "%s indices", what);
	CHECK(memcmp(a.m_PosBuffer, b.m_PosBuffer, a.m_NumVertices * 8) == 0, "%s positions", what);
	CHECK((a.m_ColorBuffer == nullptr) == (b.m_ColorBuffer == nullptr), "%s colour presence", what);
	if (a.m_ColorBuffer && b.m_ColorBuffer) { CHECK(memcmp(a.m_ColorBuffer, b.m_ColorBuffer, a.m_NumVertices * 4) == 0, "%s colours", what); }
	(void)posAliased;
}

#include <chrono>
// VGX_COMPAT_TIMING=1: the cost of one drawing (a path of 8 cubics: reset + moveTo + 8 cubicTo + close + the getters, then
// convexFillAA + polylineStrokeAA of the result) through the product's API and through the oracle, per call sequence.
template<class FN>
static double timeIt(FN fn, int reps)
{
	fn();
	const auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < reps; ++i) { fn(); }
	return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}
static void timing()
{
	bx::ShimAllocator alloc;
	vg::Path* gp = vg::createPath(&alloc); vg::Stroker* gs = vg::createStroker(&alloc);
	vgo::Path* op = vgo::createPath(&alloc); vgo::Stroker* os = vgo::createStroker(&alloc);
	if (!gp || !gs) { printf("timing: no backend\n"); return; }
	float c[8][6];
	for (int k = 0; k < 8; ++k) { const float a0 = 0.785398f * k, a1 = 0.785398f * (k + 1); c[k][0] = 100 + 60 * cosf(a0) - 20 * sinf(a0); c[k][1] = 100 + 60 * sinf(a0) + 20 * cosf(a0); c[k][2] = 100 + 60 * cosf(a1) + 20 * sinf(a1); c[k][3] = 100 + 60 * sinf(a1) - 20 * cosf(a1); c[k][4] = 100 + 60 * cosf(a1); c[k][5] = 100 + 60 * sinf(a1); }
	uint32_t sink = 0;
	const double tg = timeIt([&]() {
		vg::pathReset(gp, 1.0f, 0.25f); vg::pathMoveTo(gp, 160, 100);
		for (int k = 0; k < 8; ++k) { vg::pathCubicTo(gp, c[k][0], c[k][1], c[k][2], c[k][3], c[k][4], c[k][5]); }
		vg::pathClose(gp);
		const vg::SubPath* sp = vg::pathGetSubPaths(gp); const float* v = vg::pathGetVertices(gp);
		vg::Mesh m; vg::strokerConvexFillAA(gs, &m, v, sp[0].m_NumVertices, 0xFF112233u); sink += m.m_NumIndices;
		vg::strokerPolylineStrokeAA(gs, &m, v, sp[0].m_NumVertices, true, 0xFF445566u, 3.0f, vg::LineCap::Butt, vg::LineJoin::Miter); sink += m.m_NumIndices;
	}, 20000);
	const double to = timeIt([&]() {
		vgo::pathReset(op, 1.0f, 0.25f); vgo::pathMoveTo(op, 160, 100);
		for (int k = 0; k < 8; ++k) { vgo::pathCubicTo(op, c[k][0], c[k][1], c[k][2], c[k][3], c[k][4], c[k][5]); }
		vgo::pathClose(op);
		const vgo::SubPath* sp = vgo::pathGetSubPaths(op); const float* v = vgo::pathGetVertices(op);
		vgo::Mesh m; vgo::strokerConvexFillAA(os, &m, v, sp[0].m_NumVertices, 0xFF112233u); sink += m.m_NumIndices;
		vgo::strokerPolylineStrokeAA(os, &m, v, sp[0].m_NumVertices, true, 0xFF445566u, 3.0f, vgo::LineCap::Butt, vgo::LineJoin::Miter); sink += m.m_NumIndices;
	}, 20000);
	printf("timing: one drawing (14 API calls, %u path vertices): product %.2f us, oracle restatement %.2f us (%u)\n", vg::pathGetNumVertices(gp), tg, to, sink);
	vg::destroyStroker(gs); vg::destroyPath(gp); vgo::destroyStroker(os); vgo::destroyPath(op);
}

int main()
{
	if (getenv("VGX_COMPAT_TIMING")) { timing(); return 0; }
	bx::ShimAllocator alloc;
	// the caller's allocator (bx::AllocatorI, path.cpp:23-30 / stroker.cpp:194-200): object + host arrays must come from it
	struct Counting : public bx::ShimAllocator
	{
		long live = 0, allocs = 0;
		void* realloc(void* ptr, size_t size, size_t align, const char* f, uint32_t l) override
		{
			if (!ptr && size) { ++live; ++allocs; }
			if (ptr && !size) { --live; }
			return bx::ShimAllocator::realloc(ptr, size, align, f, l);
		}
	} counting;
	vg::Path* gp = vg::createPath(&counting);
	vg::Stroker* gs = vg::createStroker(&counting);
	if (!gp || !gs) { printf("no device\n"); return 2; }
	vgo::Path* op = vgo::createPath(&alloc);
	vgo::Stroker* os = vgo::createStroker(&alloc);

	for (int iter = 0; iter < 60; ++iter) {
		const float scale = (iter % 3 == 0) ? 2.0f : 1.0f, tol = (iter % 4 == 0) ? 0.1f : 0.25f, fringe = (iter % 5 == 0) ? 0.5f : 1.0f;
		vg::pathReset(gp, scale, tol); vgo::pathReset(op, scale, tol);
		vg::strokerReset(gs, scale, tol, fringe); vgo::strokerReset(os, scale, tol, fringe);
		const uint32_t nsub = 1 + irand(3);
		for (uint32_t s = 0; s < nsub; ++s) {
			const float sz = (irand(2) ? 20.0f : 200.0f);
			const float ox = frand(-50, 50), oy = frand(-50, 50);
			const uint32_t shape = irand(8);
			if (shape == 0) { vg::pathRect(gp, ox, oy, frand(1, sz), frand(1, sz)); vgo::pathRect(op, ox, oy, 0, 0); }
			if (shape == 0) { // keep both in sync: re-issue with identical arguments
				vgo::pathReset(op, scale, tol); vg::pathReset(gp, scale, tol);
				const float w = frand(1, sz), h = frand(1, sz);
				vg::pathRect(gp, ox, oy, w, h); vgo::pathRect(op, ox, oy, w, h);
				continue;
			}
			if (shape == 1) { const float w = frand(5, sz), h = frand(5, sz), r = frand(0, sz * 0.5f); vg::pathRoundedRect(gp, ox, oy, w, h, r); vgo::pathRoundedRect(op, ox, oy, w, h, r); continue; }
			if (shape == 2) { const float r = frand(1, sz); vg::pathCircle(gp, ox, oy, r); vgo::pathCircle(op, ox, oy, r); continue; }
			if (shape == 3) { const float w = frand(5, sz), h = frand(5, sz), a = frand(0, 10), b = frand(0, 10), c = frand(0, 10), d = frand(0, 10); vg::pathRoundedRectVarying(gp, ox, oy, w, h, a, b, c, d); vgo::pathRoundedRectVarying(op, ox, oy, w, h, a, b, c, d); continue; }
			vg::pathMoveTo(gp, ox, oy); vgo::pathMoveTo(op, ox, oy);
			const uint32_t ncmd = 1 + irand(10);
			for (uint32_t c = 0; c < ncmd; ++c) {
				const float x = ox + frand(-sz, sz), y = oy + frand(-sz, sz);
				const uint32_t t = irand(5);
				if (t == 0) { vg::pathLineTo(gp, x, y); vgo::pathLineTo(op, x, y); }
				else if (t <= 2) { const float a = frand(-sz, sz), b = frand(-sz, sz), cc = frand(-sz, sz), d = frand(-sz, sz); vg::pathCubicTo(gp, ox + a, oy + b, ox + cc, oy + d, x, y); vgo::pathCubicTo(op, ox + a, oy + b, ox + cc, oy + d, x, y); }
				else if (t == 3) { const float a = frand(-sz, sz), b = frand(-sz, sz); vg::pathQuadraticTo(gp, ox + a, oy + b, x, y); vgo::pathQuadraticTo(op, ox + a, oy + b, x, y); }
				else { const float a = frand(-sz, sz), b = frand(-sz, sz), r = frand(1, sz * 0.3f); vg::pathArcTo(gp, ox + a, oy + b, x, y, r); vgo::pathArcTo(op, ox + a, oy + b, x, y, r); }
			}
			if (irand(2)) { vg::pathClose(gp); vgo::pathClose(op); }
		}
		const uint32_t nv = vg::pathGetNumVertices(gp), nsp = vg::pathGetNumSubPaths(gp);
		CHECK(vg::vgxCompatLastStatus(gp) == 0, "path status %d", vg::vgxCompatLastStatus(gp));
		CHECK(nv == vgo::pathGetNumVertices(op) && nsp == vgo::pathGetNumSubPaths(op), "iter %d path counts %u/%u vs %u/%u", iter, nv, nsp, vgo::pathGetNumVertices(op), vgo::pathGetNumSubPaths(op));
		if (nv != vgo::pathGetNumVertices(op) || nsp != vgo::pathGetNumSubPaths(op)) { continue; }
		CHECK(memcmp(vg::pathGetVertices(gp), vgo::pathGetVertices(op), nv * 8) == 0, "iter %d path vertices", iter);
		const vg::SubPath* gsp = vg::pathGetSubPaths(gp);
		const vgo::SubPath* osp = vgo::pathGetSubPaths(op);
		for (uint32_t i = 0; i < nsp; ++i) {
			CHECK(gsp[i].m_FirstVertexID == osp[i].m_FirstVertexID && gsp[i].m_NumVertices == osp[i].m_NumVertices && gsp[i].m_IsClosed == osp[i].m_IsClosed, "iter %d sub-path %u", iter, i);
			const float* vtx = vg::pathGetVertices(gp) + 2 * gsp[i].m_FirstVertexID;
			const uint32_t n = gsp[i].m_NumVertices;
			const bool closed = gsp[i].m_IsClosed;
			const uint32_t color = 0x80000000u | g_rng;
			const float width = (irand(2) ? 3.0f : 12.0f) * scale;
			const vg::LineCap::Enum cap = (vg::LineCap::Enum)irand(3);
			const vg::LineJoin::Enum join = (vg::LineJoin::Enum)irand(3);
			vg::Mesh gm; vgo::Mesh om;
			if (n >= 2) {
				vg::strokerPolylineStrokeAA(gs, &gm, vtx, n, closed, color, width, cap, join);
				vgo::strokerPolylineStrokeAA(os, &om, vtx, n, closed, color, width, (vgo::LineCap::Enum)cap, (vgo::LineJoin::Enum)join);
				compareMesh(gm, om, false, "strokeAA");
				vg::strokerPolylineStroke(gs, &gm, vtx, n, closed, width, cap, join);
				vgo::strokerPolylineStroke(os, &om, vtx, n, closed, width, (vgo::LineCap::Enum)cap, (vgo::LineJoin::Enum)join);
				compareMesh(gm, om, false, "stroke");
				vg::strokerPolylineStrokeAAThin(gs, &gm, vtx, n, closed, color, cap, join);
				vgo::strokerPolylineStrokeAAThin(os, &om, vtx, n, closed, color, (vgo::LineCap::Enum)cap, (vgo::LineJoin::Enum)join);
				compareMesh(gm, om, false, "strokeThin");
			}
			if (n >= 3) {
				vg::strokerConvexFillAA(gs, &gm, vtx, n, color);
				vgo::strokerConvexFillAA(os, &om, vtx, n, color);
				compareMesh(gm, om, false, "fillAA");
				vg::strokerConvexFill(gs, &gm, vtx, n);
				vgo::strokerConvexFill(os, &om, vtx, n);
				compareMesh(gm, om, true, "fill");
				CHECK(gm.m_PosBuffer == vtx, "convexFill aliases the caller's vertex list");
			}
		}
	}
	// invalid stroke configuration leaves the mesh untouched (reference stroker.cpp:269-271)
	{
		const float tri[] = { 0, 0, 10, 0, 10, 10 };
		vg::Mesh m; memset(&m, 0x5A, sizeof(m)); vg::Mesh before = m;
		vg::strokerPolylineStrokeAA(gs, &m, tri, 3, false, 0xFFFFFFFFu, 4.0f, (vg::LineCap::Enum)3, vg::LineJoin::Miter);
		CHECK(memcmp(&m, &before, sizeof(m)) == 0, "invalid cap must not touch the mesh");
	}
	// grammar the reference leaves undefined: empty result + status instead of UB
	{
		vg::pathReset(gp, 1.0f, 0.25f);
		vg::pathLineTo(gp, 1, 1);
		CHECK(vg::pathGetNumVertices(gp) == 0 && vg::vgxCompatLastStatus(gp) == 2, "lineTo before moveTo");
	}
	// concave fills (stroker.h:73-85): the host's libtess2 = the one inside oracle/_ref/libvgref.so (dlopen'ed privately: that
	// library also holds the reference's own vg::strokerConcaveFill*, reached through its C wrapper vgo_concave_fill_aa)
	{
		const char* refPath = getenv("VGX_TEST_LIBVGREF");
		void* h = refPath ? dlopen(refPath, RTLD_NOW | RTLD_LOCAL) : nullptr;
		if (!h) {
			printf("concave: skipped (no oracle/_ref/libvgref.so)\n");
		} else {
			vg::VgxTessApi api;
			api.newTess = (void* (*)(void*))dlsym(h, "tessNewTess");
			api.deleteTess = (void (*)(void*))dlsym(h, "tessDeleteTess");
			api.addContour = (void (*)(void*, int, const void*, int, int))dlsym(h, "tessAddContour");
			api.tesselate = (int (*)(void*, int, int, int, int, const float*))dlsym(h, "tessTesselate");
			api.getVertexCount = (int (*)(void*))dlsym(h, "tessGetVertexCount");
			api.getVertices = (const float* (*)(void*))dlsym(h, "tessGetVertices");
			api.getElementCount = (int (*)(void*))dlsym(h, "tessGetElementCount");
			api.getElements = (const unsigned short* (*)(void*))dlsym(h, "tessGetElements");
			typedef int (*RefFn)(const float*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, float, int, float*, uint32_t*, uint16_t*, uint32_t, uint32_t, uint32_t*, uint32_t*);
			RefFn refFill = (RefFn)dlsym(h, "vgo_concave_fill_aa");
			CHECK(api.newTess && api.tesselate && refFill, "libvgref.so symbols");
			vg::vgxCompatSetTessellator(&api);
			for (int iter = 0; iter < 12 && refFill; ++iter) {
				// a star with a hole (opposite winding) and, every other time, a self-intersecting polygon
				std::vector<float> v;
				std::vector<uint32_t> first, count;
				const int pts = 5 + (int)irand(6);
				const float cx = frand(50, 500), cy = frand(50, 500), r0 = frand(40, 120), r1 = r0 * frand(0.3f, 0.7f);
				first.push_back(0); count.push_back(2 * pts);
				for (int k = 0; k < 2 * pts; ++k) { const float a = 3.14159265f * k / pts, r = (k & 1) ? r1 : r0; v.push_back(cx + r * cosf(a)); v.push_back(cy + r * sinf(a)); }
				first.push_back((uint32_t)(v.size() / 2)); count.push_back(8);
				for (int k = 0; k < 8; ++k) { const float a = -6.2831853f * k / 8, r = r1 * 0.5f; v.push_back(cx + r * cosf(a)); v.push_back(cy + r * sinf(a)); }
				if (iter & 1) {
					first.push_back((uint32_t)(v.size() / 2)); count.push_back(5);
					for (int k = 0; k < 5; ++k) { const float a = 6.2831853f * (2 * k % 5) / 5, r = r0 * 0.9f; v.push_back(cx + 30 + r * cosf(a)); v.push_back(cy - 20 + r * sinf(a)); }
				}
				const float fringe = (iter % 3 == 0) ? 0.5f : 1.0f;
				const uint32_t color = 0xC0000000u | (g_rng & 0xFFFFFFu);
				const int evenOdd = (iter >> 1) & 1;
				vg::strokerReset(gs, 1.0f, 0.25f, fringe);
				CHECK(vg::strokerConcaveFillBegin(gs), "concave begin");
				for (size_t c = 0; c < first.size(); ++c) { vg::strokerConcaveFillAddContour(gs, &v[2 * first[c]], count[c]); }
				vg::Mesh gm; memset(&gm, 0, sizeof(gm));
				CHECK(vg::strokerConcaveFillEndAA(gs, &gm, color, evenOdd ? vg::FillRule::EvenOdd : vg::FillRule::NonZero), "concave endAA status %d", vg::vgxCompatLastStatus(gs));
				std::vector<float> rp(65536 * 2); std::vector<uint32_t> rc(65536); std::vector<uint16_t> ri(65536 * 6);
				uint32_t rnv = 0, rni = 0;
				CHECK(refFill(v.data(), first.data(), count.data(), (uint32_t)first.size(), color, fringe, evenOdd, rp.data(), rc.data(), ri.data(), 65536, 65536 * 6, &rnv, &rni) == 0, "reference concave fill");
				vg::Mesh rm; rm.m_PosBuffer = rp.data(); rm.m_ColorBuffer = rc.data(); rm.m_IndexBuffer = ri.data(); rm.m_NumVertices = rnv; rm.m_NumIndices = rni;
				compareMesh(gm, rm, false, "concaveFillAA");
				// the non-AA variant is libtess2 alone: same library on both sides, must agree trivially
				CHECK(vg::strokerConcaveFillBegin(gs), "concave begin");
				for (size_t c = 0; c < first.size(); ++c) { vg::strokerConcaveFillAddContour(gs, &v[2 * first[c]], count[c]); }
				CHECK(vg::strokerConcaveFillEnd(gs, &gm, vg::FillRule::NonZero) && gm.m_NumVertices > 0 && gm.m_ColorBuffer == nullptr, "concave end");
			}
			vg::strokerConcaveFillBegin(gs); // drops the tessellator's last result before the library goes away
			vg::vgxCompatSetTessellator(nullptr);
		}
	}
	CHECK(counting.allocs > 10, "allocator used: %ld allocations", counting.allocs);
	vg::destroyStroker(gs); vg::destroyPath(gp);
	CHECK(counting.live == 0, "allocator balance after destroy: %ld live blocks", counting.live);
	{ vg::Path* p0 = vg::createPath(nullptr); vg::pathMoveTo(p0, 0, 0); vg::pathLineTo(p0, 3, 4); CHECK(vg::pathGetNumVertices(p0) == 2, "null allocator = C heap"); vg::destroyPath(p0); }
	vgo::destroyStroker(os); vgo::destroyPath(op);
	printf("%s: %d checks, %d failures\n", g_fail ? "FAILED" : "OK", g_checks, g_fail);
	return g_fail ? 1 : 0;
}
