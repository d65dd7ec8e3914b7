// vgx_example.cpp -- the C-ABI of include/vgx.h used directly from C++ (no Python, no torch): one path with a cubic
// (BASELINE config 0), filled and stroked, 1000 instances; prints the totals, the first mesh and a checksum.
//   hipcc -O2 -I include examples/vgx_example.cpp -L vg-renderer_amd -lvgx -Wl,-rpath,$PWD/vg-renderer_amd -o vgx_example
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
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

int main(int argc, char** argv)
{
	const uint64_t ninst = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000;
	vgx_ctx* ctx = nullptr;
	CHECK(vgx_create(0, &ctx));

	// moveTo(0,0) cubicTo(22.5,0, 45,22.5, 45,45)  -- what vg::moveTo / vg::cubicTo would have recorded
	const uint8_t cmdType[] = { VGX_CMD_MOVE_TO, VGX_CMD_CUBIC_TO };
	const uint32_t cmdArgOff[] = { 0, 2, 8 };
	const float args[] = { 0.0f, 0.0f, 22.5f, 0.0f, 45.0f, 22.5f, 45.0f, 45.0f };
	const uint32_t pathCmdBegin[] = { 0, 2 };
	vgx_pathset_desc desc = { cmdType, cmdArgOff, args, pathCmdBegin, 1, 2 };
	vgx_pathset* ps = nullptr;
	CHECK(vgx_pathset_create(ctx, &desc, &ps));

	// one draw per instance: strokePath(colour, width 10, Butt / Miter, AA) under a translation
	std::vector<vgx_draw> draws(ninst);
	for (uint64_t i = 0; i < ninst; ++i) {
		vgx_draw d = {};
		d.path = 0;
		d.stroke_flags = VGX_STROKE_FLAGS(VGX_CAP_BUTT, VGX_JOIN_MITER, 1, 0);
		d.stroke_color = 0xFF0000FFu;
		d.stroke_width = 10.0f;
		d.scale = 1.0f; d.tess_tol = 0.25f; d.fringe = 1.0f;
		d.mtx[0] = 1.0f; d.mtx[3] = 1.0f; d.mtx[4] = 50.0f * (float)(i % 100); d.mtx[5] = 50.0f * (float)(i / 100);
		draws[i] = d;
	}
	vgx_draw* devDraws = nullptr;
	if (hipMalloc(&devDraws, ninst * sizeof(vgx_draw)) != hipSuccess) { return 1; }
	(void)hipMemcpy(devDraws, draws.data(), ninst * sizeof(vgx_draw), hipMemcpyHostToDevice);

	vgx_sizes sz;
	CHECK(vgx_tessellate_count(ctx, ps, devDraws, ninst, &sz, nullptr));
	vgx_mesh_out out = {};
	out.cap_vertices = sz.num_vertices; out.cap_indices = sz.num_indices; out.cap_meshes = sz.num_meshes;
	(void)hipMalloc(&out.pos, sz.num_vertices * 2 * sizeof(float));
	(void)hipMalloc(&out.color, sz.num_vertices * sizeof(uint32_t));
	(void)hipMalloc(&out.idx, sz.num_indices * sizeof(uint16_t));
	(void)hipMalloc(&out.meshes, sz.num_meshes * sizeof(vgx_mesh));
	CHECK(vgx_tessellate_emit(ctx, ps, devDraws, ninst, &out, nullptr));
	(void)hipDeviceSynchronize();

	std::vector<float> pos(sz.num_vertices * 2);
	std::vector<uint16_t> idx(sz.num_indices);
	std::vector<vgx_mesh> meshes(sz.num_meshes);
	(void)hipMemcpy(pos.data(), out.pos, pos.size() * sizeof(float), hipMemcpyDeviceToHost);
	(void)hipMemcpy(idx.data(), out.idx, idx.size() * sizeof(uint16_t), hipMemcpyDeviceToHost);
	(void)hipMemcpy(meshes.data(), out.meshes, meshes.size() * sizeof(vgx_mesh), hipMemcpyDeviceToHost);
	double sum = 0.0;
	for (float v : pos) { sum += v; }
	uint64_t isum = 0;
	for (uint16_t v : idx) { isum += v; }
	printf("instances %llu  meshes %llu  vertices %llu  indices %llu  polyline vertices %llu\n", (unsigned long long)ninst,
		(unsigned long long)sz.num_meshes, (unsigned long long)sz.num_vertices, (unsigned long long)sz.num_indices, (unsigned long long)sz.num_poly_vertices);
	printf("mesh 0: %u vertices, %u indices, first position (%.6f, %.6f)\n", meshes[0].num_vertices, meshes[0].num_indices, pos[0], pos[1]);
	printf("checksum pos %.3f idx %llu\n", sum, (unsigned long long)isum);

	(void)hipFree(out.pos); (void)hipFree(out.color); (void)hipFree(out.idx); (void)hipFree(out.meshes); (void)hipFree(devDraws);
	CHECK(vgx_pathset_destroy(ctx, ps));
	CHECK(vgx_destroy(ctx));
	return 0;
}
