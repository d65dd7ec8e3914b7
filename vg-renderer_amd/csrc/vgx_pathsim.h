// vgx_pathsim.h -- exact sequential path builder (vg::Path semantics, reference src/path.cpp) for ONE lane.
// Host+device so that the builder logic can be unit-tested on the CPU (csrc/vgx_hosttest.cpp); the
// product path runs it on the device only (vgx_flatten.hip).
#ifndef VGX_PATHSIM_H
#define VGX_PATHSIM_H

#include "vgx_lane.h"
#include "vgx_internal_types.h"

// Writes the descriptor of one mesh and, when the stroker's output size is closed-form, its mesh-table
// entry. Returns true when the mesh has Round joins (caller counts those).
// mprep != null (single-pass pipeline: k_flatten_gather / k_flatten_serial): the per-mesh constants of the element kernels
// are written here too -- half widths from the stroke parameters computed below anyway, fill orientation from the first
// triangle of the mesh's (already written) polyline `poly` (stroker.cpp:721-723) -- instead of by a k_mesh_prepare pass
// that would read the descriptor and the draw record again.
VGX_HD bool vgx_write_mesh(VgxMeshDesc* mdesc, vgx_mesh* mtab, uint64_t meshIndex, const vgx_draw* dr, uint32_t drawIndex, uint32_t subIndex, uint32_t kind, bool closed, uint64_t polyFirst, uint32_t n,
	VgxMeshPrep* mprep = nullptr, const float* poly = nullptr, uint32_t orientCode = 0)
{
	VgxMeshDesc m;
	m.poly_first = polyFirst; m.poly_n = n; m.draw = drawIndex; m.subpath = subIndex;
	m.kind = kind | (closed ? 0x100u : 0u);
	uint32_t nv = VGX_MESH_NEEDS_COUNT, ni = 0;
	bool needsCount = false;
	if (kind >= VGX_MESH_STROKE) {
		const VgxStrokeParams sp = vgx_stroke_params(kind, closed, dr->stroke_flags, dr->stroke_width, dr->fringe, dr->scale, dr->tess_tol);
		m.kind |= (sp.cap << 9) | (sp.join << 11);
		const uint32_t H = (!closed && sp.cap == VGX_CAP_ROUND) ? vgx_half_circle_points(vgx_step_angle(dr->scale, sp.hsw, dr->tess_tol)) : 2u;
		needsCount = !vgx_mesh_closed_form(kind, closed, sp.cap, sp.join, n, H, &nv, &ni);
		if (mprep) {
			VgxMeshPrep pr;
			pr.f0 = sp.hsw; pr.f1 = sp.hswAA; pr.f2 = dr->fringe; pr.color = dr->stroke_color;
			mprep[meshIndex] = pr;
		}
	} else {
		vgx_mesh_closed_form(kind, closed, 0, 0, n, 2, &nv, &ni);
		if (kind == VGX_MESH_FILL_AA && (dr->fill_flags & VGX_FILL_INDEX_ORDER_SSE)) { m.kind |= 1u << 13; }
		if (mprep) {
			VgxMeshPrep pr;
			pr.f0 = 0.0f; pr.f1 = 0.0f; pr.f2 = dr->fringe; pr.color = dr->fill_color;
			if (kind == VGX_MESH_FILL_AA) { // same arithmetic as mesh_prep (vgx_elem.h)
				float sgn;
				if (orientCode & VGX_ORIENT_KNOWN) { // k_flatten_inst evaluated the first triangle while it held the vertices
					sgn = (orientCode & VGX_ORIENT_POS) ? 1.0f : ((orientCode & VGX_ORIENT_NEG) ? -1.0f : 0.0f);
				} else {
					const float* q = poly + 2 * polyFirst;
					const V2 q0 = v2(q[0], q[1]), q1 = v2(q[2], q[3]), q2 = v2(q[4], q[5]);
					sgn = vgm_sign(v2cross(v2sub(q1, q0), v2sub(q2, q0)));
				}
				pr.f0 = dr->fringe * 0.5f * sgn;
			}
			mprep[meshIndex] = pr;
		}
	}
	mdesc[meshIndex] = m;
	vgx_mesh r;
	r.first_vertex = 0; r.first_index = 0;
	r.num_vertices = needsCount ? VGX_MESH_NEEDS_COUNT : nv;
	r.num_indices = ni;
	r.draw = drawIndex;
	r.subpath_kind = (subIndex & 0x0FFFFFFFu) | ((kind & 0xFu) << 28);
	mtab[meshIndex] = r;
	return needsCount;
}

// ---- exact sequential path builder (one lane): vg::Path semantics with direct output ---------------
// Used (a) for one closed-shape command in the lane-parallel path, (b) for a whole draw in the serial
// path. Mirrors createPath..pathClose of src/path.cpp; each method cites its lines.
template<bool EMIT, bool XFORM>
struct PathSim
{
	// configuration
	float scale, tol;
	const float* mtx;        // state transform (used only when XFORM)
	float* poly;             // batch polyline array
	uint64_t polyBase;       // global index of this builder's first vertex
	vgx_subpath* subs;       // batch sub-path array (serial mode only; null in lane mode)
	uint64_t subBase;
	VgxMeshDesc* mdesc;      // serial mode only
	VgxMeshPrep* mprep;      // serial mode of the single-pass pipeline only (else null): see vgx_write_mesh
	vgx_mesh* mtab;          // serial mode only
	const vgx_draw* draw;    // serial mode only
	uint32_t numRound;       // Round-join meshes written
	uint64_t meshBase;
	uint32_t drawIndex;
	uint32_t fillFlags, strokeFlags;
	uint32_t numFillTotal;   // emit/serial: fill meshes of the draw (stroke meshes come after them)
	// state
	uint32_t nverts;         // path.cpp:10 m_NumVertices (relative to polyBase)
	uint32_t nsubs;
	uint32_t nfill, nstroke; // mesh ranks so far
	bool open;               // a current sub-path exists
	uint32_t spFirst, spN;
	bool spClosed;
	V2 first, last;          // untransformed first / last vertex of the current sub-path
	uint32_t limit;          // emit: vertices [0, limit) are stored. limit = the FINAL vertex count (known from the
	                         // count pass), so the vertex pathClose pops at the very end is never written; earlier
	                         // popped slots are simply overwritten by the next sub-path's first vertex (same lane).
	// lane-mode results
	bool laneExists, laneClosed;

	VGX_HDM void init()
	{
		nverts = 0; nsubs = 0; nfill = 0; nstroke = 0; numRound = 0; open = false; spFirst = 0; spN = 0; spClosed = false;
		first = v2(0.0f, 0.0f); last = first; laneExists = false; laneClosed = false;
	}
	VGX_HDM void store(uint32_t i, V2 p)
	{
		if (EMIT) {
			if (XFORM) { p = v2xform(p, mtx); }
			float* o = poly + 2 * (polyBase + i);
			o[0] = p.x;
			o[1] = p.y;
		}
	}
	VGX_HDM void endSub() // sub-path is complete: write its record and its mesh descriptors
	{
		if (!open) { return; }
		const uint32_t subIndex = nsubs - 1;
		if (EMIT && subs) {
			vgx_subpath r;
			r.first_vertex = polyBase + spFirst;
			r.num_vertices = spN;
			r.flags = spClosed ? 1u : 0u;
			subs[subBase + subIndex] = r;
		}
		if ((fillFlags & VGX_FILL_ENABLE) && spN >= 3) {
			if (EMIT && mdesc) {
				vgx_write_mesh(mdesc, mtab, meshBase + nfill, draw, drawIndex, subIndex, (fillFlags & VGX_FILL_AA) ? VGX_MESH_FILL_AA : VGX_MESH_FILL, spClosed, polyBase + spFirst, spN, mprep, poly);
			}
			++nfill;
		}
		if ((strokeFlags & VGX_STROKE_ENABLE) && spN >= 2) {
			if (EMIT && mdesc) {
				const uint32_t k = !(strokeFlags & VGX_STROKE_AA) ? VGX_MESH_STROKE : ((strokeFlags & VGX_STROKE_THIN) ? VGX_MESH_STROKE_AA_THIN : VGX_MESH_STROKE_AA);
				if (vgx_write_mesh(mdesc, mtab, meshBase + numFillTotal + nstroke, draw, drawIndex, subIndex, k, spClosed, polyBase + spFirst, spN, mprep, poly)) { ++numRound; }
			}
			++nstroke;
		}
		open = false;
	}
	VGX_HDM void raw(float x, float y) // pathAllocVertices + write, no dedup (path.cpp:748-759)
	{
		last = v2(x, y);
		if (nverts < limit) { store(nverts, last); }
		++nverts;
		++spN;
	}
	VGX_HDM void add(float x, float y) // pathAddVertex, path.cpp:761-784
	{
		if (spN != 0 && v2near(last, v2(x, y))) {
			return;
		}
		if (spN == 0) { first = v2(x, y); }
		raw(x, y);
	}
	VGX_HDM void moveTo(float x, float y) // path.cpp:62-78
	{
		if (!open || spN != 0) {
			endSub();
			open = true;
			laneExists = true;
			spFirst = nverts; spN = 0; spClosed = false;
			++nsubs;
		}
		add(x, y);
	}
	VGX_HDM void lineTo(float x, float y) { add(x, y); } // path.cpp:80-84
	VGX_HDM void close() // path.cpp:707-726
	{
		if (spClosed || spN <= 2) {
			return;
		}
		spClosed = true;
		laneClosed = true;
		if (v2near(last, first)) {
			--spN;
			--nverts;
		}
	}
	// sink interface of vgx_flatten_cubic
	VGX_HDM void leaf(float x, float y) { add(x, y); }
	VGX_HDM void dropped() {}

	template<class STACK>
	VGX_HDM void cubicTo(float c1x, float c1y, float c2x, float c2y, float x, float y, STACK& st) // path.cpp:86-182
	{
		const float tessTol = tol / (scale * scale);
		vgx_flatten_cubic(last.x, last.y, c1x, c1y, c2x, c2y, x, y, tessTol, st, *this);
	}
	template<class STACK>
	VGX_HDM void quadTo(float cx, float cy, float x, float y, STACK& st)
	{
		float c1x, c1y, c2x, c2y;
		vgx_quad_to_cubic(last.x, last.y, cx, cy, x, y, &c1x, &c1y, &c2x, &c2y);
		cubicTo(c1x, c1y, c2x, c2y, x, y, st);
	}
	// rotation recurrence of every arc writer (path.cpp:322-336, 609-628, 669-681)
	VGX_HDM void rotated(float cx, float cy, float rx, float ry, float ca, float sa, float cosD, float sinD, uint32_t count)
	{
		for (uint32_t i = 0; i < count; ++i) {
			const float ns = sinD * ca + cosD * sa;
			const float nc = cosD * ca - sinD * sa;
			ca = nc;
			sa = ns;
			if (spN == 0) { first = v2(cx + rx * ca, cy + ry * sa); }
			raw(cx + rx * ca, cy + ry * sa);
		}
	}
	VGX_HDM void rect(float x, float y, float w, float h) // path.cpp:275-286
	{
		if (vgm_abs(w) < VGM_EPSILON || vgm_abs(h) < VGM_EPSILON) {
			return;
		}
		moveTo(x, y);
		lineTo(x, y + h);
		lineTo(x + w, y + h);
		lineTo(x + w, y);
		close();
	}
	VGX_HDM void ellipse(float cx, float cy, float rx, float ry) // path.cpp:599-631
	{
		const float avgR = (rx + ry) * 0.5f;
		const float da = vgx_step_angle(scale, avgR, tol);
		const uint32_t numPoints = vgx_half_circle_points(da) * 2;
		moveTo(cx + rx, cy);
		const float dtheta = -VGM_PI2 / (float)numPoints;
		rotated(cx, cy, rx, ry, 1.0f, 0.0f, vgm_cos(dtheta), vgm_sin(dtheta), numPoints - 1);
		close();
	}
	VGX_HDM void roundedRect(float x, float y, float w, float h, float r) // path.cpp:288-409
	{
		if (r < 0.1f) {
			rect(x, y, w, h);
			return;
		}
		const float maxR = vgm_min(w, h) * 0.5f;
		if (w == h && r >= maxR - VGM_EPSILON) {
			ellipse(x + maxR, y + maxR, maxR, maxR);
			return;
		}
		r = vgm_min(r, maxR);
		const float da = vgx_step_angle(scale, r, tol);
		const uint32_t quarter = (vgx_half_circle_points(da) >> 1) + 1;
		const float dtheta = -VGM_PIHALF / (float)(quarter - 1);
		const float cosD = vgm_cos(dtheta);
		const float sinD = vgm_sin(dtheta);
		moveTo(x, y + r);
		lineTo(x, y + h - r);
		rotated(x + r, y + h - r, r, r, -1.0f, 0.0f, cosD, sinD, quarter - 1);
		lineTo(x + w - r, y + h);
		rotated(x + w - r, y + h - r, r, r, 0.0f, 1.0f, cosD, sinD, quarter - 1);
		lineTo(x + w, y + r);
		rotated(x + w - r, y + r, r, r, 1.0f, 0.0f, cosD, sinD, quarter - 1);
		lineTo(x + r, y);
		rotated(x + r, y + r, r, r, 0.0f, -1.0f, cosD, sinD, quarter - 1);
		close();
	}
	VGX_HDM void variedCorner(float rc, float cx, float cy, float ca, float sa) // path.cpp:428-455
	{
		const float halfDa = vgm_acos((scale * rc) / ((scale * rc) + tol));
		const uint32_t half = vgm_umax(2u, vgx_point_count(vgm_ceil(VGM_PIHALF / halfDa)));
		const uint32_t quarter = (half >> 1) + 1;
		const float dtheta = -VGM_PIHALF / (float)(quarter - 1);
		rotated(cx, cy, rc, rc, ca, sa, vgm_cos(dtheta), vgm_sin(dtheta), quarter - 1);
	}
	VGX_HDM void roundedRectVarying(float x, float y, float w, float h, float rTL, float rTR, float rBR, float rBL) // path.cpp:411-559
	{
		if (rTL < 0.1f && rBL < 0.1f && rBR < 0.1f && rTR < 0.1f) {
			rect(x, y, w, h);
			return;
		}
		const float halfw = w * 0.5f;
		const float halfh = h * 0.5f;
		const float rtl = vgm_min(vgm_min(rTL, halfw), halfh);
		const float rtr = vgm_min(vgm_min(rTR, halfw), halfh);
		const float rbl = vgm_min(vgm_min(rBL, halfw), halfh);
		const float rbr = vgm_min(vgm_min(rBR, halfw), halfh);
		if (rtl < 0.1f) { moveTo(x, y); } else { moveTo(x + rtl, y); variedCorner(rtl, x + rtl, y + rtl, 0.0f, -1.0f); }
		if (rbl < 0.1f) { lineTo(x, y + h); } else { lineTo(x, y + h - rbl); variedCorner(rbl, x + rbl, y + h - rbl, -1.0f, 0.0f); }
		if (rbr < 0.1f) { lineTo(x + w, y + h); } else { lineTo(x + w - rbr, y + h); variedCorner(rbr, x + w - rbr, y + h - rbr, 0.0f, 1.0f); }
		if (rtr < 0.1f) { lineTo(x + w, y); } else { lineTo(x + w, y + rtr); variedCorner(rtr, x + w - rtr, y + rtr, 1.0f, 0.0f); }
		close();
	}
	VGX_HDM void arc(float cx, float cy, float r, float a0, float a1, bool cw) // path.cpp:633-682
	{
		while (a0 > VGM_PI2) { a0 -= VGM_PI2; }
		while (a1 > VGM_PI2) { a1 -= VGM_PI2; }
		if (!cw) {
			while (a0 < a1) { a0 += VGM_PI2; }
		} else {
			while (a1 < a0) { a1 += VGM_PI2; }
		}
		const float da = vgx_step_angle(scale, r, tol);
		const uint32_t numPoints = vgm_umax(2u, vgx_point_count(vgm_ceil(vgm_abs(a1 - a0) / da)));
		const float dtheta = (a1 - a0) / (float)numPoints;
		const float cosD = vgm_cos(dtheta);
		const float sinD = vgm_sin(dtheta);
		const float ca = vgm_cos(a0);
		const float sa = vgm_sin(a0);
		if (open && spN != 0) { lineTo(cx + r * ca, cy + r * sa); } else { moveTo(cx + r * ca, cy + r * sa); }
		rotated(cx, cy, r, r, ca, sa, cosD, sinD, numPoints);
	}
	VGX_HDM void arcTo(float x1, float y1, float x2, float y2, float r) // path.cpp:203-273
	{
		float dx0 = last.x - x1, dy0 = last.y - y1;
		float dx1 = x2 - x1, dy1 = y2 - y1;
		{
			const float lenSqr = dx0 * dx0 + dy0 * dy0;
			const float invLen = lenSqr < VGM_EPSILON ? 0.0f : vgm_rsqrt(lenSqr);
			dx0 *= invLen; dy0 *= invLen;
		}
		{
			const float lenSqr = dx1 * dx1 + dy1 * dy1;
			const float invLen = lenSqr < VGM_EPSILON ? 0.0f : vgm_rsqrt(lenSqr);
			dx1 *= invLen; dy1 *= invLen;
		}
		const float a = vgm_acos(dx0 * dx1 + dy0 * dy1);
		const float d = r / vgm_tan(a / 2.0f);
		if (d > 10000.0f) {
			lineTo(x1, y1);
			return;
		}
		const float crs = dx1 * dy0 - dx0 * dy1;
		if (crs > 0.0f) {
			arc(x1 + dx0 * d + dy0 * r, y1 + dy0 * d - dx0 * r, r, vgm_atan2(dx0, -dy0), vgm_atan2(-dx1, dy1), true);
		} else {
			arc(x1 + dx0 * d - dy0 * r, y1 + dy0 * d + dx0 * r, r, vgm_atan2(-dx0, dy0), vgm_atan2(dx1, -dy1), false);
		}
	}
	VGX_HDM void polyline(const float* coords, uint32_t numPoints) // path.cpp:684-705
	{
		if (spN > 0 && numPoints > 0 && v2near(last, v2(coords[0], coords[1]))) {
			coords += 2;
			--numPoints;
		}
		for (uint32_t i = 0; i < numPoints; ++i) {
			if (spN == 0) { first = v2(coords[2 * i], coords[2 * i + 1]); }
			raw(coords[2 * i], coords[2 * i + 1]);
		}
	}
	// one closed-shape command
	VGX_HDM void shape(uint32_t type, const float* a)
	{
		switch (type) {
		case VGX_CMD_RECT: rect(a[0], a[1], a[2], a[3]); break;
		case VGX_CMD_ROUNDED_RECT: roundedRect(a[0], a[1], a[2], a[3], a[4]); break;
		case VGX_CMD_ROUNDED_RECT_VARYING: roundedRectVarying(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
		case VGX_CMD_CIRCLE: ellipse(a[0], a[1], a[2], a[2]); break;
		case VGX_CMD_ELLIPSE: ellipse(a[0], a[1], a[2], a[3]); break;
		default: break;
		}
	}
	// a whole draw, command after command (serial path)
	template<class STACK>
	VGX_HDM void run(const VgxPathSetDev& ps, uint32_t c0, uint32_t c1, STACK& st)
	{
		for (uint32_t c = c0; c < c1; ++c) {
			const float* a = ps.args + ps.cmd_arg_off[c];
			const uint32_t na = ps.cmd_arg_off[c + 1] - ps.cmd_arg_off[c];
			const uint32_t type = ps.cmd_type[c];
			switch (type) {
			case VGX_CMD_MOVE_TO: moveTo(a[0], a[1]); break;
			case VGX_CMD_LINE_TO: lineTo(a[0], a[1]); break;
			case VGX_CMD_CUBIC_TO: cubicTo(a[0], a[1], a[2], a[3], a[4], a[5], st); break;
			case VGX_CMD_QUAD_TO: quadTo(a[0], a[1], a[2], a[3], st); break;
			case VGX_CMD_CLOSE: close(); break;
			case VGX_CMD_ARC_TO: arcTo(a[0], a[1], a[2], a[3], a[4]); break;
			case VGX_CMD_ARC: arc(a[0], a[1], a[2], a[3], a[4], a[5] != 0.0f); break;
			case VGX_CMD_POLYLINE: polyline(a, na >> 1); break;
			default: shape(type, a); break;
			}
		}
		endSub();
	}
};


#endif
