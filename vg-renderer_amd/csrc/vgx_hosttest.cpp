// vgx_hosttest.cpp -- CPU build of the lane-level product code (vgx_lane.h, vgx_pathsim.h) for unit tests.
// This is NOT a fallback path: nothing in the package loads it; tests/test_host_lane_logic.py uses it to
// check the per-lane builder logic against the oracle without a GPU.
#include "vgx_pathsim.h"
#include "vgx_inst.h"
#include "vgx_pathset_host.h"
#include "vgx_thin.h"
#include <string.h>

namespace {
struct HostStack
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
}

extern "C" {

// Runs the exact serial builder (the device's slow path) for ONE draw on the host.
// counts[4] = { num_poly_vertices, num_subpaths, num_fill_meshes, num_stroke_meshes }
int vgxt_serial_flatten(const vgx_pathset_desc* d, const vgx_draw* draw, int applyTransform, float* poly, vgx_subpath* subs, uint32_t* counts)
{
	VgxPathSetDev ps;
	memset(&ps, 0, sizeof(ps));
	ps.cmd_type = d->cmd_type; ps.cmd_arg_off = d->cmd_arg_off; ps.args = d->args; ps.path_cmd_begin = d->path_cmd_begin;
	ps.npaths = d->npaths; ps.ncmd = d->ncmd;
	HostStack st;
	const uint32_t c0 = d->path_cmd_begin[draw->path], c1 = d->path_cmd_begin[draw->path + 1];
	uint32_t limit = 0;
	if (poly) { // the emit pass knows the final vertex count from the count pass
		uint32_t tmp[4];
		vgxt_serial_flatten(d, draw, 0, nullptr, nullptr, tmp);
		limit = tmp[0];
	}
	if (poly && applyTransform) {
		PathSim<true, true> sim;
		sim.scale = draw->scale; sim.tol = draw->tess_tol; sim.mtx = draw->mtx; sim.poly = poly; sim.polyBase = 0;
		sim.subs = subs; sim.subBase = 0; sim.mdesc = nullptr; sim.mprep = nullptr; sim.mtab = nullptr; sim.draw = nullptr; sim.limit = limit; sim.meshBase = 0; sim.drawIndex = 0;
		sim.fillFlags = draw->fill_flags; sim.strokeFlags = draw->stroke_flags; sim.numFillTotal = 0;
		sim.init();
		sim.run(ps, c0, c1, st);
		counts[0] = sim.nverts; counts[1] = sim.nsubs; counts[2] = sim.nfill; counts[3] = sim.nstroke;
	} else if (poly) {
		PathSim<true, false> sim;
		sim.scale = draw->scale; sim.tol = draw->tess_tol; sim.mtx = draw->mtx; sim.poly = poly; sim.polyBase = 0;
		sim.subs = subs; sim.subBase = 0; sim.mdesc = nullptr; sim.mprep = nullptr; sim.mtab = nullptr; sim.draw = nullptr; sim.limit = limit; sim.meshBase = 0; sim.drawIndex = 0;
		sim.fillFlags = draw->fill_flags; sim.strokeFlags = draw->stroke_flags; sim.numFillTotal = 0;
		sim.init();
		sim.run(ps, c0, c1, st);
		counts[0] = sim.nverts; counts[1] = sim.nsubs; counts[2] = sim.nfill; counts[3] = sim.nstroke;
	} else {
		PathSim<false, false> sim;
		sim.scale = draw->scale; sim.tol = draw->tess_tol; sim.mtx = nullptr; sim.poly = nullptr; sim.polyBase = 0;
		sim.subs = nullptr; sim.subBase = 0; sim.mdesc = nullptr; sim.mprep = nullptr; sim.mtab = nullptr; sim.draw = nullptr; sim.limit = limit; sim.meshBase = 0; sim.drawIndex = 0;
		sim.fillFlags = draw->fill_flags; sim.strokeFlags = draw->stroke_flags; sim.numFillTotal = 0;
		sim.init();
		sim.run(ps, c0, c1, st);
		counts[0] = sim.nverts; counts[1] = sim.nsubs; counts[2] = sim.nfill; counts[3] = sim.nstroke;
	}
	return 0;
}

// One lane of the instanced flatten kernel (InstCore, vgx_inst.h) for ONE draw on the host: the command loop of
// k_flatten_inst with the generic cubic walk, a bump allocator over `heap` (cap vertices, lane blocks of `lb` vertices) in
// place of the wave-aggregated one. `cursor` carries the heap position from draw to draw (lane-private blocks persist in
// the kernel; here every call starts a fresh block). sub_rec[k] is written at the sub-path-ending commands (k relative to
// the path's first command); counts[5] = { poly vertices, sub-paths, fill meshes, stroke meshes, 1 if the heap ran out }.
namespace {
struct HostInstEnv
{
	float* poly; uint64_t cap; uint32_t lb; uint64_t* cursor; bool failed;
	bool alloc(uint64_t want, uint64_t* base)
	{
		if (*cursor + want > cap) { failed = true; return false; }
		*base = *cursor; *cursor += want;
		return true;
	}
	void emit(float* wp, float x, float y) { wp[0] = x; wp[1] = y; }
	void flushForMove(float*) {}
};
}
int vgxt_inst_flatten(const vgx_pathset_desc* d, const vgx_draw* draw, float* heap, uint64_t cap, uint32_t lb, uint64_t* cursor, VgxSubRec* sub_rec, uint32_t* counts)
{
	HostStack st;
	InstCore<HostInstEnv> L;
	L.env.poly = heap; L.env.cap = cap; L.env.lb = lb; L.env.cursor = cursor; L.env.failed = false;
	L.initLane();
	L.beginDraw(draw->mtx, draw->scale, draw->tess_tol, draw->fill_flags, draw->stroke_flags);
	const uint32_t c0 = d->path_cmd_begin[draw->path], c1 = d->path_cmd_begin[draw->path + 1];
	for (uint32_t c = c0; c < c1; ++c) {
		const float* a = d->args + d->cmd_arg_off[c];
		const uint32_t na = d->cmd_arg_off[c + 1] - d->cmd_arg_off[c];
		const uint32_t type = d->cmd_type[c];
		switch (type) {
		case VGX_CMD_MOVE_TO: L.moveTo(a[0], a[1]); break;
		case VGX_CMD_LINE_TO: L.lineTo(a[0], a[1]); break;
		case VGX_CMD_CUBIC_TO: L.cubicTo(a[0], a[1], a[2], a[3], a[4], a[5], st); break;
		case VGX_CMD_QUAD_TO: L.quadTo(a[0], a[1], a[2], a[3], st); break;
		case VGX_CMD_CLOSE: L.close(); break;
		case VGX_CMD_POLYLINE: L.polyline(a, na >> 1); break;
		default: return -1; // statically serial paths never reach the instanced lane
		}
		const bool lastInSub = (c + 1 == c1) || d->cmd_type[c + 1] == VGX_CMD_MOVE_TO;
		if (lastInSub) { L.endSub(sub_rec + (c - c0)); }
	}
	const vgx_draw_info di = L.drawInfo();
	counts[0] = di.num_poly_vertices; counts[1] = di.num_subpaths; counts[2] = di.flags >> 1; counts[3] = di.num_meshes - (di.flags >> 1);
	counts[4] = L.env.failed ? 1u : 0u;
	return 0;
}

// closed-form mesh sizes (vgx_lane.h) for one mesh of a draw; returns 1 when the size is closed-form
int vgxt_mesh_closed_form(const vgx_draw* dr, uint32_t kind, int closed, uint32_t n, uint32_t* nv, uint32_t* ni)
{
	if (kind >= VGX_MESH_STROKE) {
		const VgxStrokeParams sp = vgx_stroke_params(kind, closed != 0, dr->stroke_flags, dr->stroke_width, dr->fringe, dr->scale, dr->tess_tol);
		const uint32_t H = vgx_half_circle_points(vgx_step_angle(dr->scale, sp.hsw, dr->tess_tol));
		return vgx_mesh_closed_form(kind, closed != 0, sp.cap, sp.join, n, H, nv, ni) ? 1 : 0;
	}
	return vgx_mesh_closed_form(kind, closed != 0, 0, 0, n, 2, nv, ni) ? 1 : 0;
}

// the pinned transcendentals (csrc/vgmath.h), for tests that restate arithmetic in numpy
float vgxt_cos(float a) { return vgm_cos(a); }
float vgxt_sin(float a) { return vgm_sin(a); }
// vectors of them (tests/test_oracle_libm.py: ULP distance of every pinned transcendental from glibc's):
// fn 0 cos, 1 sin, 2 tan, 3 acos, 4 atan2(a, b), 5 rsqrt
void vgxt_math_vec(int fn, const float* a, const float* b, float* out, uint64_t n)
{
	for (uint64_t i = 0; i < n; ++i) {
		switch (fn) {
		case 0: out[i] = vgm_cos(a[i]); break;
		case 1: out[i] = vgm_sin(a[i]); break;
		case 2: out[i] = vgm_tan(a[i]); break;
		case 3: out[i] = vgm_acos(a[i]); break;
		case 4: out[i] = vgm_atan2(a[i], b[i]); break;
		default: out[i] = vgm_rsqrt(a[i]); break;
		}
	}
}

} // extern "C"

extern "C" {

// Path sets of MOVE_TO / LINE_TO / CLOSE paths (vgx_thin.h): the static layout tables as vgx_pathset_create builds them, then
// k_flatten_thin's lane function over every command instance of the batch, one after the other. Outputs as the kernel leaves
// them: poly [command instances][2] (vertex v of draw d at cmd_prefix[d] + v), sub_rec [static sub-paths of the batch] (record j of
// draw d at sub_prefix[d] + j), dinfo [ndraws], serial [ndraws] (1: the draw is listed for the exact builder).
// Returns 1, 0 when the set is not eligible, < 0 = -(validation error).
int vgxt_thin_flatten(const vgx_pathset_desc* d, const vgx_draw* draws, uint64_t ndraws, float* poly, VgxSubRec* subRec, vgx_draw_info* dinfo, uint8_t* serial)
{
	std::vector<uint8_t> cmdFlags, pathFlags;
	std::vector<uint32_t> spStart;
	uint32_t maxCmds = 0;
	const int st = vgx_pathset_validate_host(d, &cmdFlags, &spStart, &pathFlags, &maxCmds);
	if (st != VGX_OK) { return -st; }
	std::vector<uint32_t> pathSubBegin(d->npaths + 1, 0);
	uint32_t nsub = 0;
	for (uint32_t p = 0; p < d->npaths; ++p) {
		pathSubBegin[p] = nsub;
		for (uint32_t c = d->path_cmd_begin[p]; c < d->path_cmd_begin[p + 1]; ++c) { if (cmdFlags[c] & VGX_CF_LAST_IN_SUB) { ++nsub; } }
	}
	pathSubBegin[d->npaths] = nsub;
	std::vector<VgxCmdThin> thv(d->ncmd + 3);
	VgxCmdThin* th = thv.data() + 1;
	std::vector<VgxThinPath> tp(d->npaths + 1);
	std::vector<VgxThinSub> ts(nsub + 1);
	vgx_thin_fill(d, cmdFlags.data(), spStart.data(), pathFlags.data(), th);
	if (d->npaths == 0 || d->ncmd == 0 || !vgx_thin_build(d->npaths, d->path_cmd_begin, pathFlags.data(), pathSubBegin.data(), th, tp.data(), ts.data())) { return 0; }
	uint64_t cmdPrefix = 0, subPrefix = 0;
	for (uint64_t i = 0; i < ndraws; ++i) {
		const vgx_draw* dr = draws + i;
		const VgxThinPath q = tp[dr->path];
		const uint32_t ncmd = d->path_cmd_begin[dr->path + 1] - d->path_cmd_begin[dr->path];
		serial[i] = 0;
		for (uint32_t k = 0; k < ncmd; ++k) {
			if (vgx_thin_lane(q, th[q.pc0 + k], ts.data(), dr->mtx, dr->fill_flags, dr->stroke_flags, cmdPrefix, &subPrefix, poly, subRec, dinfo + i)) { serial[i] = 1; }
		}
		cmdPrefix += ncmd;
		subPrefix += pathSubBegin[dr->path + 1] - pathSubBegin[dr->path];
	}
	return 1;
}

}

extern "C" {

// The derived tables of a path set as the host loops build them (vgx_pathset_host.h, vgx_thin.h): the oracle of the device-side
// build of vgx_pathset_create (vgx_pathset.hip). `which` = VGX_PS_TABLE_* (include/vgx.h). Returns the table's size in bytes
// (dst may be NULL to ask), < 0 = -(validation status).
int64_t vgxt_pathset_table(const vgx_pathset_desc* d, int which, void* dst, uint64_t cap)
{
	std::vector<uint8_t> cmdFlags, pathFlags;
	std::vector<uint32_t> spStart, pathSubBegin, subLastCmd;
	uint32_t maxCmds = 0;
	const int st = vgx_pathset_validate_host(d, &cmdFlags, &spStart, &pathFlags, &maxCmds);
	if (st != VGX_OK) { return -(int64_t)st; }
	vgx_pathset_subs_host(d, cmdFlags.data(), &pathSubBegin, &subLastCmd);
	std::vector<VgxCmdThin> thv(d->ncmd + 3);
	memset(thv.data(), 0, thv.size() * sizeof(VgxCmdThin));
	VgxCmdThin* th = thv.data() + 1;
	std::vector<VgxThinPath> tp(d->npaths + 1);
	std::vector<VgxThinSub> ts(subLastCmd.size() + 1);
	vgx_thin_fill(d, cmdFlags.data(), spStart.data(), pathFlags.data(), th);
	const bool thinStatic = d->npaths != 0 && d->ncmd != 0 && vgx_thin_build(d->npaths, d->path_cmd_begin, pathFlags.data(), pathSubBegin.data(), th, tp.data(), ts.data());
	bool hasSerial = false, hasEmpty = false;
	for (uint32_t i = 0; i < d->npaths; ++i) {
		if (pathFlags[i] & VGX_PF_SERIAL) { hasSerial = true; }
		if (d->path_cmd_begin[i + 1] == d->path_cmd_begin[i]) { hasEmpty = true; }
	}
	std::vector<VgxCmdRec> rec;
	const void* src = nullptr; uint64_t n = 0;
	uint32_t scal[8] = { maxCmds, hasSerial ? 1u : 0u, hasEmpty ? 1u : 0u, thinStatic ? 1u : 0u, (uint32_t)subLastCmd.size(), d->npaths, d->ncmd, 0u };
	switch (which) {
	case VGX_PS_TABLE_CMD_FLAGS: src = cmdFlags.data(); n = d->ncmd; break;
	case VGX_PS_TABLE_SP_START: src = spStart.data(); n = (uint64_t)d->ncmd * 4; break;
	case VGX_PS_TABLE_PATH_FLAGS: src = pathFlags.data(); n = d->npaths; break;
	case VGX_PS_TABLE_CMDREC:
		rec.resize(d->ncmd + 1);
		vgx_pathset_records_host(d, cmdFlags.data(), spStart.data(), rec.data());
		src = rec.data(); n = (uint64_t)d->ncmd * sizeof(VgxCmdRec); break;
	case VGX_PS_TABLE_PATH_SUB_BEGIN: src = pathSubBegin.data(); n = ((uint64_t)d->npaths + 1) * 4; break;
	case VGX_PS_TABLE_SUB_LAST_CMD: src = subLastCmd.data(); n = (uint64_t)subLastCmd.size() * 4; break;
	case VGX_PS_TABLE_CMDTHIN: src = th; n = (uint64_t)d->ncmd * sizeof(VgxCmdThin); break;
	case VGX_PS_TABLE_THIN_PATH: src = tp.data(); n = thinStatic ? (uint64_t)d->npaths * sizeof(VgxThinPath) : 0; break;
	case VGX_PS_TABLE_THIN_SUB: src = ts.data(); n = thinStatic ? (uint64_t)subLastCmd.size() * sizeof(VgxThinSub) : 0; break;
	case VGX_PS_TABLE_SCALARS: src = scal; n = sizeof(scal); break;
	default: return -(int64_t)VGX_E_INVALID_ARG;
	}
	if (dst) {
		if (cap < n) { return -(int64_t)VGX_E_NOSPACE; }
		if (n) { memcpy(dst, src, n); }
	}
	return (int64_t)n;
}

}
