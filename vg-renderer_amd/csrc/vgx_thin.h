// vgx_thin.h -- path sets of MOVE_TO / LINE_TO / CLOSE paths only: the polyline LAYOUT of such a path is static.
//
// pathMoveTo / pathLineTo / pathClose (path.cpp:64-85, 707-726) decide everything in local space: which commands add a vertex,
// which vertex pathClose removes, where sub-paths begin and whether they are closed do not depend on the draw. k_flatten_build
// finds that out again for every command instance of every batch (two wave scans, a dozen ballots / shuffles and the carries of
// the segmented bookkeeping: ~3 400 wave instructions per 64 commands, 0.24 ms for BASELINE configs[3]'s 10 M lineTo commands
// whose vertices a copy kernel moves in 0.05 ms). Here the decisions are taken ONCE, when the path set is created
// (vgx_thin_build, host), with the same tests in the same order as the kernel's lanes:
//   per command  the index of its vertex inside the path's polyline (VgxCmdThin::pad; none for CLOSE and for the vertex a
//                closing pathClose pops), the ordinal of its sub-path (VgxCmdThin::meta bits 16-31)
//   per sub-path first vertex (relative to the path), vertex count | closed << 31
//   per path     vertices, sub-paths, sub-paths of >= 3 / >= 2 vertices (the draw's fill / stroke meshes), DEGENERATE (a
//                lineTo closer than epsilon to its start point: pathLineTo drops it, path.cpp:769-775 -- the exact one-lane
//                builder does such draws, as it does for k_flatten_build)
// and k_flatten_thin (vgx_flatten.hip) is gather - transform - scatter: one lane per command instance, vertex k of draw d at
// poly[cmd_prefix[d] + k] (a path has at most one vertex per command, so the command prefix places the draws: no scan, no heap
// blocks, only sub-paths have to be contiguous), the sub-path records and the draw's counts from the tables.
// vgx_thin_lane is what one lane does; host and device compile the same function (tests/test_host_lane_logic.py runs it on the
// CPU against the oracle, csrc/vgx_hosttest.cpp).
#ifndef VGX_THIN_H
#define VGX_THIN_H

#include "vgx_lane.h"
#include "vgx_internal_types.h"

#define VGX_THIN_NONE 0xFFFFFFFFu
#define VGX_THIN_DEGENERATE 1u

struct VgxThinPath // 32 bytes
{
	uint32_t pc0;     // first command of the path
	uint32_t nverts;  // polyline vertices of the path
	uint32_t nsubs;   // sub-paths (MOVE_TO commands)
	uint32_t nge3;    // sub-paths of >= 3 vertices: fill meshes of a draw that fills
	uint32_t nge2;    // ... of >= 2 vertices: stroke meshes of a draw that strokes
	uint32_t flags;   // VGX_THIN_DEGENERATE
	uint32_t sub0;    // first entry of the path in the VgxThinSub table (= path_sub_begin)
	uint32_t pad;
};

struct VgxThinSub // 8 bytes
{
	uint32_t first;   // first vertex, relative to the path's first vertex
	uint32_t info;    // vertex count | closed << 31 (VgxSubRec::info)
};

// Host: which paths are thin (VGX_PF_THIN into pathFlags) and the thin records of every command: meta = type | flags << 8, the
// command's point (CLOSE: the first point of its sub-path, the MOVE_TO that spStart names).
static inline void vgx_thin_fill(const vgx_pathset_desc* desc, const uint8_t* cmdFlags, const uint32_t* spStart, uint8_t* pathFlags, VgxCmdThin* th)
{
	for (uint32_t p = 0; p < desc->npaths; ++p) {
		bool thin = !(pathFlags[p] & VGX_PF_SERIAL) && desc->path_cmd_begin[p + 1] > desc->path_cmd_begin[p];
		for (uint32_t c = desc->path_cmd_begin[p]; thin && c < desc->path_cmd_begin[p + 1]; ++c) {
			const uint32_t t = desc->cmd_type[c];
			if (t != VGX_CMD_MOVE_TO && t != VGX_CMD_LINE_TO && t != VGX_CMD_CLOSE) { thin = false; }
		}
		if (thin) { pathFlags[p] |= VGX_PF_THIN; }
	}
	for (uint32_t c = 0; c < desc->ncmd; ++c) {
		const uint32_t t = desc->cmd_type[c];
		const uint32_t ao = desc->cmd_arg_off[c];
		th[c].meta = t | ((uint32_t)cmdFlags[c] << 8);
		th[c].x = 0.0f; th[c].y = 0.0f; th[c].pad = 0;
		if (t == VGX_CMD_MOVE_TO || t == VGX_CMD_LINE_TO) { th[c].x = desc->args[ao]; th[c].y = desc->args[ao + 1]; }
		else if (t == VGX_CMD_CLOSE) {
			const uint32_t hc = spStart[c];
			if (desc->cmd_type[hc] == VGX_CMD_MOVE_TO) { const uint32_t ho = desc->cmd_arg_off[hc]; th[c].x = desc->args[ho]; th[c].y = desc->args[ho + 1]; }
		}
	}
}

// Host: the tables of every path of the set. th[c] holds meta = type | flags << 8 and the points (vgx_pathset_create); pad and
// the upper half of meta are written here. Returns false when the set is not eligible (a path that is not thin, or one with
// more than 65 535 sub-paths): the caller keeps k_flatten_build for the set.
static inline bool vgx_thin_build(uint32_t npaths, const uint32_t* pathCmdBegin, const uint8_t* pathFlags, const uint32_t* pathSubBegin,
                                  VgxCmdThin* th, VgxThinPath* tp, VgxThinSub* ts)
{
	for (uint32_t p = 0; p < npaths; ++p) {
		if (!(pathFlags[p] & VGX_PF_THIN)) { return false; }
	}
	for (uint32_t p = 0; p < npaths; ++p) {
		const uint32_t c0 = pathCmdBegin[p], c1 = pathCmdBegin[p + 1];
		VgxThinPath q;
		q.pc0 = c0; q.nverts = 0; q.nsubs = 0; q.nge3 = 0; q.nge2 = 0; q.flags = 0; q.sub0 = pathSubBegin[p]; q.pad = 0;
		uint32_t nv = 0;       // vertices of the path so far (the kernel's exclusive scan, pops included)
		uint32_t sp = 0;       // vertices of the open sub-path so far
		uint32_t subFirst = 0; // its first vertex
		for (uint32_t c = c0; c < c1; ++c) {
			const uint32_t type = th[c].meta & 0xFFu, fl = (th[c].meta >> 8) & 0xFFu;
			if (fl & VGX_CF_STARTS_SUB) { sp = 0; subFirst = nv; }
			int cnt = 0;
			bool closedHere = false;
			th[c].pad = VGX_THIN_NONE;
			if (type == VGX_CMD_MOVE_TO) {
				cnt = 1; q.nsubs++;
			} else if (type == VGX_CMD_LINE_TO) {
				cnt = 1;
				if (v2near(v2(th[c - 1].x, th[c - 1].y), v2(th[c].x, th[c].y))) { q.flags |= VGX_THIN_DEGENERATE; }
			} else { // CLOSE: th[c].x / y = the sub-path's first point, the record in front = the last vertex
				if (sp > 2) { // pathClose, path.cpp:707-726
					closedHere = true;
					if (v2near(v2(th[c - 1].x, th[c - 1].y), v2(th[c].x, th[c].y))) {
						cnt = -1;
						th[c - 1].pad = VGX_THIN_NONE; // the vertex in front is removed: never stored, its place is the next vertex's
					}
				}
			}
			if (cnt == 1) { th[c].pad = nv; }
			if (q.nsubs == 0 || q.nsubs > 65536u) { return false; } // (validated paths start with MOVE_TO)
			th[c].meta = (th[c].meta & 0xFFFFu) | ((q.nsubs - 1u) << 16);
			const uint32_t spTotal = (uint32_t)((int)sp + cnt);
			if (fl & VGX_CF_LAST_IN_SUB) {
				VgxThinSub s;
				s.first = subFirst; s.info = spTotal | (closedHere ? 0x80000000u : 0u);
				ts[q.sub0 + q.nsubs - 1u] = s;
				if (spTotal >= 3u) { q.nge3++; }
				if (spTotal >= 2u) { q.nge2++; }
			}
			nv = (uint32_t)((int)nv + cnt);
			sp = spTotal;
		}
		q.nverts = nv;
		if (q.nsubs != pathSubBegin[p + 1] - pathSubBegin[p]) { return false; } // (one LAST_IN_SUB command per MOVE_TO: the tables would not line up)
		tp[p] = q;
	}
	return true;
}

// One lane = command k of draw d (a path of the tables above). polyBase = cmd_prefix[d], subBase = &sub_prefix[d] (read by the
// lanes that end a sub-path only); mtx / fillFlags / strokeFlags: the draw's.
// Returns true when the draw belongs on the serial list (its last lane says so, once).
VGX_HD bool vgx_thin_lane(const VgxThinPath& q, const VgxCmdThin& t, const VgxThinSub* ts, const float* mtx, uint32_t fillFlags, uint32_t strokeFlags,
                          uint64_t polyBase, const uint64_t* subBase, float* poly, VgxSubRec* sub_rec, vgx_draw_info* di_out)
{
	const uint32_t fl = (t.meta >> 8) & 0xFFu;
	const bool degenerate = (q.flags & VGX_THIN_DEGENERATE) != 0;
	if (!degenerate) {
		if (t.pad != VGX_THIN_NONE) {
			const V2 p = v2xform(v2(t.x, t.y), mtx); // transformPos2D, vg_util.h:24-28
			float* out = poly + 2 * (polyBase + (uint64_t)t.pad);
#if defined(__HIP_DEVICE_COMPILE__)
			*(float2*)out = make_float2(p.x, p.y);
#else
			out[0] = p.x; out[1] = p.y;
#endif
		}
		if (fl & VGX_CF_LAST_IN_SUB) {
			const uint32_t j = t.meta >> 16;
			const VgxThinSub s = ts[q.sub0 + j];
			VgxSubRec sr;
			sr.first = polyBase + (uint64_t)s.first; sr.info = s.info; sr.pad = 0;
			sub_rec[*subBase + (uint64_t)j] = sr;
		}
	}
	if (fl & VGX_CF_LAST_IN_PATH) {
		vgx_draw_info di;
		di.first_poly_vertex = polyBase; di.first_subpath = 0; di.first_mesh = 0;
		if (degenerate) {
			di.num_poly_vertices = 0; di.num_subpaths = 0; di.num_meshes = 0; di.flags = 1u;
		} else {
			const uint32_t nf = (fillFlags & VGX_FILL_ENABLE) ? q.nge3 : 0u;
			const uint32_t ns = (strokeFlags & VGX_STROKE_ENABLE) ? q.nge2 : 0u;
			di.num_poly_vertices = q.nverts; di.num_subpaths = q.nsubs; di.num_meshes = nf + ns; di.flags = nf << 1;
		}
		*di_out = di;
		return degenerate;
	}
	return false;
}

#endif
