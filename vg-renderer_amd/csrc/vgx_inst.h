// vgx_inst.h -- the per-lane sequential path builder of the INSTANCED flatten kernel (vgx_inst.hip).
//
// A batch that draws the same sequence of paths again and again (draws[i].path == draws[i mod P].path: one drawing
// submitted for many instances, the shape of the headline workload) is flattened with ONE LANE PER INSTANCE: the 64 lanes
// of a wave walk the same path for 64 different draw records. Every lane runs vg::Path's own sequential algorithm
// (pathMoveTo / pathLineTo / pathCubicTo / pathClose / pathPolyline, reference src/path.cpp:62-201, 684-784) -- the
// epsilon de-duplication of pathAddVertex and the pop of pathClose happen in order, so there is no "degenerate draw"
// escape hatch on this path -- and because all lanes hold the same cubic, the depth-first subdivision that diverges
// when neighbouring lanes hold different commands (31 % lane use in k_flatten_build) runs in lock-step.
//
// InstCore is that lane. It is __host__ __device__ so the builder logic -- sub-path bookkeeping and the lane-private
// heap blocks with their sub-path moves -- is unit-tested on the CPU against the oracle (csrc/vgx_hosttest.cpp,
// tests/test_host_lane_logic.py); the product only runs it on the device, where ENV::alloc is a wave-aggregated bump of
// the polyline heap and the cubic walk is the hand-shaped loop of vgx_inst.hip.
//
// ENV: poly / cap / lb (heap, its capacity, vertices per block), alloc(want, &base) (bump allocation), emit(wp, x, y)
// (vertex store; the device stages 8 vertices per lane in LDS and writes whole 64-byte pieces), flushForMove(wp)
// (everything emitted so far must be readable at its heap address).
//
// Heap: a lane appends its vertices to a lane-private block of the polyline heap. Only a sub-path has to be contiguous:
// when the block is full the lane takes a fresh one (room for twice the current sub-path's vertices, at least ENV::lb)
// and moves the vertices its current sub-path already has. Space bound (vgx_tessellate_count): blocks left behind waste
// less than one sub-path each (< the useful vertices of that block when sub-paths are <= lb / 2 long), longer sub-paths
// grow geometrically (<= 4x their size + one block), plus every lane's last open block.
#ifndef VGX_INST_H
#define VGX_INST_H

#include "vgx_lane.h"
#include "vgx_internal_types.h"

#define VGX_INST_BLOCK 256u       /* polyline vertices per lane-private heap block (default) */
#define VGX_INST_LONG_SUBPATH 128u /* = VGX_INST_BLOCK / 2: sub-paths longer than this are sized as geometric growers */
#define VGX_INST_WAVES 4096        /* persistent one-wave workgroups; 64 lanes each keep one open block */
#define VGX_INST_MIN_INSTANCES 32u /* fewer repetitions than this: the command-parallel kernel is the better mapping */

template<class ENV>
struct InstCore
{
	ENV env;
	// the draw (one lane = one draw record)
	float m0, m1, m2, m3, m4, m5; // state transform
	float tessTol;                // tess_tol / scale^2 (path.cpp:104)
	uint32_t fillFlags, strokeFlags;
	// the lane's place in the polyline heap
	float* wp;        // next vertex goes here
	uint32_t room;    // free vertices at wp
	uint64_t spFirst; // heap index of the current sub-path's first vertex
	// current sub-path (vg::SubPath, include/vg/path.h:11-16)
	uint32_t spN;
	bool spClosed;
	bool dead;        // the heap is exhausted (grow): the write position no longer follows the vertex count
	V2 first, last;   // untransformed first / last vertex of the current sub-path
	V2 q0, q1, q2;    // TRANSFORMED first three vertices of the current sub-path (fill orientation, endSub)
	// per-draw counters
	uint32_t nverts, nsubs, nfill, nstroke;
	uint64_t drawFirst;

	VGX_HDM void initLane()
	{
		wp = env.poly; room = 0; spFirst = 0; spN = 0; spClosed = false; dead = false;
		first = v2(0.0f, 0.0f); last = first; q0 = first; q1 = first; q2 = first;
		nverts = 0; nsubs = 0; nfill = 0; nstroke = 0; drawFirst = 0;
	}
	VGX_HDM void beginDraw(const float* mtx, float scale, float tol, uint32_t ff, uint32_t sf)
	{
		m0 = mtx[0]; m1 = mtx[1]; m2 = mtx[2]; m3 = mtx[3]; m4 = mtx[4]; m5 = mtx[5];
		tessTol = tol / (scale * scale);
		fillFlags = ff; strokeFlags = sf;
		nverts = 0; nsubs = 0; nfill = 0; nstroke = 0;
		spN = 0; spClosed = false;
		drawFirst = spFirst;
	}
	// The lane's block is full: continue in a fresh one, the current sub-path's vertices move along.
	VGX_HDM void grow()
	{
		uint64_t want = 2ull * spN + 2ull;
		if (want < env.lb) { want = env.lb; }
		want = (want + 7ull) & ~7ull; // blocks are whole 64-byte pieces (the device stages vertices per piece, ENV::emit)
		uint64_t base = 0;
		if (!env.alloc(want, &base)) {
			// heap exhausted (status = VGX_E_NOSPACE is set, every later kernel exits at once): keep the lane inside the
			// heap -- it overwrites other lanes' dead data from the start of the heap -- until its loops end
			wp = env.poly;
			room = env.cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (env.cap ? (uint32_t)env.cap : 1u);
			spFirst = 0;
			dead = true;
			return;
		}
		float* dst = env.poly + 2 * base;
		const float* src = wp - 2 * (uint64_t)spN;
		env.flushForMove(wp); // the tail of the sub-path may still be staged
		for (uint32_t i = 0; i < spN; ++i) {
			env.emit(dst + 2 * (uint64_t)i, src[2 * (uint64_t)i], src[2 * (uint64_t)i + 1]);
		}
		wp = dst + 2 * (uint64_t)spN;
		const uint64_t left = want - spN;
		room = left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left;
		spFirst = base;
	}
	// pathAllocVertices(1) + write (path.cpp:748-759, 779-783), with transformPath's arithmetic applied on the way out
	// (vg_util.h:24-28: (m0 * x + m2 * y) + m4)
	// vertex number spN of the sub-path is (ox, oy): the first three are kept for the orientation test of the fill mesh
	VGX_HDM void note(float ox, float oy)
	{
		if (spN == 0) { q0 = v2(ox, oy); } else if (spN == 1) { q1 = v2(ox, oy); } else if (spN == 2) { q2 = v2(ox, oy); }
	}
	VGX_HDM void put(V2 p)
	{
		if (room == 0) { grow(); }
		const float ox = m0 * p.x + m2 * p.y + m4, oy = m1 * p.x + m3 * p.y + m5;
		if (spN < 3) { note(ox, oy); }
		env.emit(wp, ox, oy);
		wp += 2;
		--room;
		++spN;
		last = p;
	}
	VGX_HDM void add(V2 p) // pathAddVertex, path.cpp:761-784
	{
		if (spN != 0 && v2near(last, p)) { return; }
		if (spN == 0) { first = p; }
		put(p);
	}
	VGX_HDM void moveTo(float x, float y) // path.cpp:62-78: every MOVE_TO opens a sub-path (the previous one has >= 1 vertex)
	{
		spN = 0; spClosed = false;
		add(v2(x, y));
	}
	VGX_HDM void lineTo(float x, float y) { add(v2(x, y)); } // path.cpp:80-84
	VGX_HDM void close() // path.cpp:707-726
	{
		if (spClosed || spN <= 2) { return; }
		spClosed = true;
		if (v2near(last, first)) { // the last vertex is dropped; its slot is reused by whatever comes next
			--spN;
			if (!dead) { wp -= 2; ++room; }
		}
	}
	VGX_HDM void polyline(const float* coords, uint32_t numPoints) // path.cpp:684-705: only the first point is tested
	{
		if (spN > 0 && numPoints > 0 && v2near(last, v2(coords[0], coords[1]))) {
			coords += 2;
			--numPoints;
		}
		for (uint32_t i = 0; i < numPoints; ++i) {
			const V2 p = v2(coords[2 * i], coords[2 * i + 1]);
			if (spN == 0) { first = p; }
			put(p);
		}
	}
	// sink interface of vgx_flatten_cubic (vgx_lane.h)
	VGX_HDM void leaf(float x, float y) { add(v2(x, y)); }
	VGX_HDM void dropped() {}
	template<class STACK>
	VGX_HDM void cubicTo(float c1x, float c1y, float c2x, float c2y, float x, float y, STACK& st) // path.cpp:86-182
	{
		vgx_flatten_cubic(last.x, last.y, c1x, c1y, c2x, c2y, x, y, tessTol, st, *this);
	}
	template<class STACK>
	VGX_HDM void quadTo(float cx, float cy, float x, float y, STACK& st) // path.cpp:184-201
	{
		float c1x, c1y, c2x, c2y;
		vgx_quad_to_cubic(last.x, last.y, cx, cy, x, y, &c1x, &c1y, &c2x, &c2y);
		cubicTo(c1x, c1y, c2x, c2y, x, y, st);
	}
	// The command just processed was the last one of its sub-path: write the record k_flatten_gather turns into mesh
	// descriptors (same record, same place as k_flatten_build: the command instance of that last command).
	VGX_HDM void endSub(VgxSubRec* rec)
	{
		VgxSubRec sr;
		sr.first = spFirst; sr.info = spN | (spClosed ? 0x80000000u : 0u);
		// sign of the first triangle, evaluated as mesh_prep / vgx_write_mesh do (stroker.cpp:721-723): k_flatten_gather then
		// does not have to fetch three vertices per fill mesh from the heap (0.11 of its 0.48 ms)
		const float orient = v2cross(v2sub(q1, q0), v2sub(q2, q0));
		sr.pad = VGX_ORIENT_KNOWN | (0.0f < orient ? VGX_ORIENT_POS : 0u) | (0.0f > orient ? VGX_ORIENT_NEG : 0u);
		*rec = sr;
		if ((fillFlags & VGX_FILL_ENABLE) && spN >= 3) { ++nfill; }     // vg.cpp:3099-3131
		if ((strokeFlags & VGX_STROKE_ENABLE) && spN >= 2) { ++nstroke; } // vg.cpp:3448-3485
		nverts += spN;
		++nsubs;
		spFirst += spN;
		spN = 0;
		spClosed = false;
	}
	VGX_HDM vgx_draw_info drawInfo() const
	{
		vgx_draw_info di;
		di.first_poly_vertex = drawFirst; di.first_subpath = 0; di.first_mesh = 0;
		di.num_poly_vertices = nverts; di.num_subpaths = nsubs; di.num_meshes = nfill + nstroke;
		di.flags = nfill << 1;
		return di;
	}
};

#endif // VGX_INST_H
