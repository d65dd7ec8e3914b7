// vgx_hosttest.cpp -- CPU build of the lane-level product code (vgx_lane.h, vgx_pathsim.h) for unit tests.
// This is NOT a fallback path: nothing in the package loads it; tests/test_host_lane_logic.py uses it to
// check the per-lane builder logic against the oracle without a GPU.
#include "vgx_pathsim.h"
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
