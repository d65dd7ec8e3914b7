// vgx_static_scene.cpp -- a retained scene under a moving camera through the C-ABI (no Python, no torch): what a caller of the reference
// does with createCommandList once + submitCommandList every frame (src/vg.cpp:4332-4625). 400 distinct paths (polygons and open
// polylines), each filled and / or stroked with Round joins, drawn once each: no period, so without the caller's promise every frame runs
// flatten + scans + fill + stroke. With vgx_set_static_batches the count flattens the list once and a frame is the template kernels alone;
// a structural change (here: two draws swapped) comes back as VGX_E_STALE and is answered by counting again.
//   hipcc -O2 -I include examples/vgx_static_scene.cpp -L vg-renderer_amd -lvgx -Wl,-rpath,$PWD/vg-renderer_amd -o vgx_static_scene
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "vgx.h"

#define CHECK(call)                                                                        \
	do {                                                                                   \
		const int st_ = (call);                                                            \
		if (st_ != VGX_OK) {                                                               \
			fprintf(stderr, "%s failed: %s (%d)\n", #call, vgx_status_string(st_), st_); \
			return 1;                                                                      \
		}                                                                                  \
	} while (0)

static uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main(int argc, char** argv)
{
	const int frames = argc > 1 ? atoi(argv[1]) : 50;
	vgx_ctx* ctx = nullptr;
	CHECK(vgx_create(0, &ctx));

	// the scene: every path a ring of 5-40 points around its own centre; two out of three closed
	const uint32_t npaths = 400;
	std::vector<uint8_t> cmdType;
	std::vector<uint32_t> cmdArgOff(1, 0u), pathCmdBegin(1, 0u);
	std::vector<float> args;
	uint32_t seed = 12345u;
	for (uint32_t p = 0; p < npaths; ++p) {
		const uint32_t n = 5 + rnd(seed) % 36;
		const float cx = (float)(rnd(seed) % 1200), cy = (float)(rnd(seed) % 700), r = 10.0f + (float)(rnd(seed) % 60);
		for (uint32_t k = 0; k < n; ++k) {
			const float a = 6.2831853f * (float)k / (float)n, rr = r * (0.6f + 0.4f * (float)(rnd(seed) % 100) / 100.0f);
			cmdType.push_back(k == 0 ? VGX_CMD_MOVE_TO : VGX_CMD_LINE_TO);
			args.push_back(cx + rr * cosf(a)); args.push_back(cy + rr * sinf(a));
			cmdArgOff.push_back((uint32_t)args.size());
		}
		if (p % 3 != 0) { cmdType.push_back(VGX_CMD_CLOSE); cmdArgOff.push_back((uint32_t)args.size()); }
		pathCmdBegin.push_back((uint32_t)cmdType.size());
	}
	vgx_pathset_desc desc = { cmdType.data(), cmdArgOff.data(), args.data(), pathCmdBegin.data(), npaths, (uint32_t)cmdType.size() };
	vgx_pathset* ps = nullptr;
	CHECK(vgx_pathset_create(ctx, &desc, &ps));

	// one draw per path, a style of its own each (what fillPath / strokePath under one State would have recorded)
	std::vector<vgx_draw> draws(npaths);
	for (uint32_t p = 0; p < npaths; ++p) {
		vgx_draw d;
		memset(&d, 0, sizeof(d));
		d.path = p;
		if (p % 3 != 0) { d.fill_flags = VGX_FILL_ENABLE | VGX_FILL_AA; d.fill_color = 0xFF000000u | rnd(seed); }
		d.stroke_flags = VGX_STROKE_FLAGS(p % 3 == 0 ? VGX_CAP_ROUND : VGX_CAP_BUTT, VGX_JOIN_ROUND, 1, 0);
		d.stroke_color = 0xFF000000u | rnd(seed);
		d.stroke_width = 2.0f + (float)(p % 5);
		d.scale = 1.0f; d.tess_tol = 0.25f; d.fringe = 1.0f;
		d.mtx[0] = 1.0f; d.mtx[3] = 1.0f;
		draws[p] = d;
	}
	vgx_draw* devDraws = nullptr;
	if (hipMalloc(&devDraws, npaths * sizeof(vgx_draw)) != hipSuccess) { return 1; }
	(void)hipMemcpy(devDraws, draws.data(), npaths * sizeof(vgx_draw), hipMemcpyHostToDevice);

	CHECK(vgx_set_static_batches(ctx, 1)); // the promise: between two counts only transforms / colours move
	vgx_sizes sz;
	CHECK(vgx_tessellate_count(ctx, ps, devDraws, npaths, &sz, nullptr));
	// Round joins: the sizes follow the transform. Room for the zoomed-in frames as well (a frame that outgrows it says VGX_E_NOSPACE)
	vgx_mesh_out out;
	memset(&out, 0, sizeof(out));
	out.cap_vertices = sz.num_vertices * 2; out.cap_indices = sz.num_indices * 2; out.cap_meshes = sz.num_meshes;
	(void)hipMalloc(&out.pos, out.cap_vertices * 2 * sizeof(float));
	(void)hipMalloc(&out.color, out.cap_vertices * sizeof(uint32_t));
	(void)hipMalloc(&out.idx, out.cap_indices * sizeof(uint16_t));
	(void)hipMalloc(&out.meshes, out.cap_meshes * sizeof(vgx_mesh));
	vgx_sizes* devSizes = nullptr; uint32_t* devStatus = nullptr;
	(void)hipMalloc(&devSizes, sizeof(vgx_sizes)); (void)hipMalloc(&devStatus, sizeof(uint32_t));
	printf("scene: %u paths, %u commands -> %llu meshes, %llu vertices, %llu indices at the counted camera\n", npaths, (unsigned)cmdType.size(),
		(unsigned long long)sz.num_meshes, (unsigned long long)sz.num_vertices, (unsigned long long)sz.num_indices);

	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	float usTotal = 0.0f;
	unsigned long long vmin = ~0ull, vmax = 0;
	for (int f = 0; f < frames; ++f) {
		// the camera: rotate and zoom about the canvas centre (the draws' scale stays the scene's: a caller that lets avgScale follow the
		// zoom changes tolerances and stroke widths, i.e. the structure -- VGX_E_STALE, count again)
		const float a = 0.02f * (float)f, z = 1.0f + 0.2f * sinf(0.3f * (float)f), c = z * cosf(a), s = z * sinf(a);
		for (uint32_t p = 0; p < npaths; ++p) {
			draws[p].mtx[0] = c; draws[p].mtx[1] = s; draws[p].mtx[2] = -s; draws[p].mtx[3] = c;
			draws[p].mtx[4] = 600.0f - (c * 600.0f - s * 350.0f); draws[p].mtx[5] = 350.0f - (s * 600.0f + c * 350.0f);
		}
		(void)hipMemcpyAsync(devDraws, draws.data(), npaths * sizeof(vgx_draw), hipMemcpyHostToDevice, nullptr);
		(void)hipEventRecord(e0, nullptr);
		CHECK(vgx_tessellate(ctx, ps, devDraws, npaths, &out, devSizes, devStatus, nullptr));
		(void)hipEventRecord(e1, nullptr);
		uint32_t status = 0; vgx_sizes got;
		(void)hipMemcpy(&status, devStatus, sizeof(status), hipMemcpyDeviceToHost);
		(void)hipMemcpy(&got, devSizes, sizeof(got), hipMemcpyDeviceToHost);
		if (status != VGX_OK) { fprintf(stderr, "frame %d: %s\n", f, vgx_status_string((int)status)); return 1; }
		float ms = 0.0f;
		(void)hipEventElapsedTime(&ms, e0, e1);
		if (f >= 5) { usTotal += ms * 1000.0f; }
		vmin = got.num_vertices < vmin ? got.num_vertices : vmin; vmax = got.num_vertices > vmax ? got.num_vertices : vmax;
	}
	printf("%d frames as a static batch: %.1f us per frame on the device; vertices per frame %llu .. %llu (counted per frame: Round joins depend on the transformed geometry; a rotation + uniform zoom keeps them)\n",
		frames, frames > 5 ? usTotal / (float)(frames - 5) : 0.0f, vmin, vmax);

	// a structural change: two draws of different paths swapped
	vgx_draw t = draws[7]; draws[7] = draws[311]; draws[311] = t;
	(void)hipMemcpy(devDraws, draws.data(), npaths * sizeof(vgx_draw), hipMemcpyHostToDevice);
	CHECK(vgx_tessellate(ctx, ps, devDraws, npaths, &out, devSizes, devStatus, nullptr));
	uint32_t status = 0;
	(void)hipMemcpy(&status, devStatus, sizeof(status), hipMemcpyDeviceToHost);
	printf("after swapping two draws: %s\n", vgx_status_string((int)status));
	if (status != VGX_E_STALE) { return 1; }
	CHECK(vgx_tessellate_count(ctx, ps, devDraws, npaths, &sz, nullptr)); // the answer: count again
	CHECK(vgx_tessellate(ctx, ps, devDraws, npaths, &out, devSizes, devStatus, nullptr));
	(void)hipMemcpy(&status, devStatus, sizeof(status), hipMemcpyDeviceToHost);
	printf("after counting again: %s\n", vgx_status_string((int)status));
	if (status != VGX_OK) { return 1; }

	vgx_pathset_destroy(ctx, ps);
	vgx_destroy(ctx);
	return 0;
}
