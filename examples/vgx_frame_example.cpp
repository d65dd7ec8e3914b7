// vgx_frame_example.cpp -- one frame from a recorded vg::CommandList to bgfx-ready buffers, in plain C++ on the C-ABI:
//   bytes of a command list (written here by hand in the reference's wire format, vg.cpp:243-247, 2403-2690)
//   -> vgx_cmdlist_decode (host: paths + draws + per-draw state)          [replaces ctxSubmitCommandList's interpreter]
//   -> vgx_pathset_create, vgx_tessellate_count, vgx_tessellate (device): the meshes of the path draws
//   -> vgx_set_assembly + vgx_merge_uv (device): the user meshes of the list's IndexedTriList commands (vg::indexedTriList,
//      vg.cpp:4129-4175; the decoder hands them over) put at their draws' places, the frame assembled
//   -> vertex streams (pos / uv / colour), ONE index buffer rebased per draw command, the draw-command table
//      (what createDrawCommand_VertexColor / allocDrawCommand leave for vg::end to upload, vg.cpp:5207-5460, 1076-1288).
//   hipcc -O2 -I include examples/vgx_frame_example.cpp -L vg-renderer_amd -lvgx -Wl,-rpath,$PWD/vg-renderer_amd -o vgx_frame_example
#include <hip/hip_runtime.h>
#include <stdio.h>
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

// vg::CommandType values used here (vg.cpp:177-241) and the record layout: CommandHeader{uint32 type, uint32 size}
// padded to 16 bytes, then the payload padded to 16 bytes (clAllocCommand, vg.cpp:5694-5723)
enum { CT_BeginPath = 0, CT_MoveTo = 1, CT_LineTo = 2, CT_CubicTo = 3, CT_Rect = 7, CT_Circle = 10, CT_ClosePath = 13, CT_FillPathColor = 14,
       CT_StrokePathColor = 17, CT_IndexedTriList = 20, CT_PushState = 28, CT_PopState = 29, CT_SetScissor = 31, CT_TransformTranslate = 35 };
struct ListWriter
{
	std::vector<uint8_t> b;
	void cmd(uint32_t type, const void* payload, uint32_t n)
	{
		const uint32_t padded = (n + 15u) & ~15u;
		const uint32_t hdr[4] = { type, padded, 0, 0 };
		b.insert(b.end(), (const uint8_t*)hdr, (const uint8_t*)hdr + 16);
		b.insert(b.end(), (const uint8_t*)payload, (const uint8_t*)payload + n);
		b.insert(b.end(), padded - n, 0);
	}
	void f(uint32_t type, std::initializer_list<float> v) { std::vector<float> a(v); cmd(type, a.data(), (uint32_t)a.size() * 4); }
	void fill(uint32_t color, bool aa) { const uint32_t p[2] = { aa ? 4u : 0u, color }; cmd(CT_FillPathColor, p, 8); }           // VG_FILL_FLAGS
	// clIndexedTriList (vg.cpp:2566-2611): uint32 nv, float2 pos[nv], uint32 nuv, int16x2 uv[nuv], uint32 nc, Color col[nc], uint32 ni, uint16 idx[ni], uint16 image
	void triList(const float* pos, const int16_t* uv, uint32_t nv, const uint32_t* col, uint32_t nc, const uint16_t* idx, uint32_t ni, uint16_t image)
	{
		std::vector<uint8_t> p;
		auto put = [&](const void* src, size_t n) { p.insert(p.end(), (const uint8_t*)src, (const uint8_t*)src + n); };
		const uint32_t nuv = uv ? nv : 0u;
		put(&nv, 4); put(pos, (size_t)nv * 8); put(&nuv, 4); if (uv) { put(uv, (size_t)nv * 4); }
		put(&nc, 4); put(col, (size_t)nc * 4); put(&ni, 4); put(idx, (size_t)ni * 2); put(&image, 2);
		cmd(CT_IndexedTriList, p.data(), (uint32_t)p.size());
	}
	void stroke(uint32_t color, float w, uint32_t cap, uint32_t join, bool aa)                                                       // VG_STROKE_FLAGS
	{
		uint8_t p[12]; const uint32_t flags = ((aa ? 1u : 0u) << 4) | (cap << 2) | join;
		memcpy(p, &w, 4); memcpy(p + 4, &flags, 4); memcpy(p + 8, &color, 4);
		cmd(CT_StrokePathColor, p, 12);
	}
};

int main()
{
	// ---- a small drawing, as vg::clBeginPath / clRect / clFillPath ... would have recorded it ----
	ListWriter L;
	for (int i = 0; i < 300; ++i) {
		const float x = 20.0f + 40.0f * (float)(i % 30), y = 20.0f + 60.0f * (float)(i / 30);
		L.cmd(CT_BeginPath, nullptr, 0);
		if (i % 3 == 0) { L.f(CT_Rect, { x, y, 30.0f, 40.0f }); }
		else if (i % 3 == 1) { L.f(CT_Circle, { x + 15.0f, y + 20.0f, 14.0f }); }
		else { L.f(CT_MoveTo, { x, y }); L.f(CT_CubicTo, { x + 30.0f, y, x + 30.0f, y + 40.0f, x, y + 40.0f }); L.cmd(CT_ClosePath, nullptr, 0); }
		L.fill(0xFF2060C0u + (uint32_t)i, true);
		if (i % 2) { L.stroke(0xFF000000u, 2.0f, 0, 0, true); }
		if (i == 150) { L.f(CT_SetScissor, { 0.0f, 0.0f, 640.0f, 720.0f }); } // a scissor change: a new draw command from here on
		if (i % 100 == 50) { // a user mesh between the paths: a textured quad on image 3 with its own UVs and per-vertex colours
			const float q[8] = { x, y, x + 30.0f, y, x + 30.0f, y + 30.0f, x, y + 30.0f };
			const int16_t uv[8] = { 0, 0, 32767, 0, 32767, 32767, 0, 32767 };
			const uint32_t col[4] = { 0xFFFFFFFFu, 0xFF0000FFu, 0xFF00FF00u, 0xFFFF0000u };
			const uint16_t idx[6] = { 0, 1, 2, 0, 2, 3 };
			L.triList(q, uv, 4, col, 4, idx, 6, 3);
		}
	}

	// ---- host: decode (count pass, then store pass) ----
	vgx_cmdlist_state st = {};
	st.mtx[0] = 1.0f; st.mtx[3] = 1.0f; st.global_alpha = 1.0f; st.tess_tol = 0.25f; st.fringe = 1.0f;
	st.canvas_width = 1280.0f; st.canvas_height = 720.0f;
	vgx_cmdlist_out o = {};
	CHECK(vgx_cmdlist_decode(L.b.data(), (uint32_t)L.b.size(), &st, &o));
	std::vector<uint8_t> cmdType(o.num_cmds + 1);
	std::vector<uint32_t> argOff(o.num_cmds + 1), pathBegin(o.num_paths + 1);
	std::vector<float> args(o.num_args + 1);
	std::vector<vgx_draw> draws(o.num_draws + 1);
	std::vector<vgx_draw_state> dstate(o.num_draws + 1);
	o.cmd_type = cmdType.data(); o.cmd_arg_off = argOff.data(); o.args = args.data(); o.path_cmd_begin = pathBegin.data();
	o.draws = draws.data(); o.draw_state = dstate.data();
	o.cap_cmds = o.num_cmds; o.cap_args = o.num_args; o.cap_paths = o.num_paths; o.cap_draws = o.num_draws;
	// the list's user meshes (IndexedTriList): positions through the state transform, colours, UVs, mesh-local indices, one record each
	std::vector<float> triPos(o.num_tri_vertices * 2 + 2);
	std::vector<uint32_t> triCol(o.num_tri_vertices + 1), triUV(o.num_tri_vertices + 1);
	std::vector<uint16_t> triIdx(o.num_tri_indices + 1);
	std::vector<vgx_mesh> triMesh(o.num_tri_meshes + 1);
	o.tri_pos = triPos.data(); o.tri_color = triCol.data(); o.tri_uv = triUV.data(); o.tri_idx = triIdx.data(); o.tri_meshes = triMesh.data();
	o.cap_tri_vertices = o.num_tri_vertices; o.cap_tri_indices = o.num_tri_indices; o.cap_tri_meshes = o.num_tri_meshes;
	st.white_uv[0] = 0x003F003Fu; st.font_image = 0; // getWhitePixelUV / m_FontImages[0] of the Context the list is submitted to
	CHECK(vgx_cmdlist_decode(L.b.data(), (uint32_t)L.b.size(), &st, &o));
	printf("list: %zu bytes -> %u paths, %u path commands, %u draws (%u of them user meshes), %u skipped\n", L.b.size(), o.num_paths, o.num_cmds, o.num_draws, o.num_tri_meshes, o.num_skipped);

	// ---- device: tessellate with draw-command assembly armed ----
	vgx_ctx* ctx = nullptr;
	CHECK(vgx_create(0, &ctx));
	vgx_pathset_desc desc = { cmdType.data(), argOff.data(), args.data(), pathBegin.data(), o.num_paths, o.num_cmds };
	vgx_pathset* ps = nullptr;
	CHECK(vgx_pathset_create(ctx, &desc, &ps));
	vgx_draw* devDraws = nullptr;
	if (hipMalloc(&devDraws, o.num_draws * sizeof(vgx_draw)) != hipSuccess) { return 1; }
	(void)hipMemcpy(devDraws, draws.data(), o.num_draws * sizeof(vgx_draw), hipMemcpyHostToDevice);
	vgx_sizes sz;
	CHECK(vgx_tessellate_count(ctx, ps, devDraws, o.num_draws, &sz, nullptr));

	const uint32_t maxVB = 4096; // Config::m_MaxVBVertices: small, so that the frame needs several vertex buffers
	// sequence A: the meshes of the path draws (the user-mesh draws have none here)
	vgx_mesh_out seqA = {};
	seqA.cap_vertices = sz.num_vertices; seqA.cap_indices = sz.num_indices; seqA.cap_meshes = sz.num_meshes;
	(void)hipMalloc(&seqA.pos, (sz.num_vertices + 1) * 2 * sizeof(float));
	(void)hipMalloc(&seqA.color, (sz.num_vertices + 1) * sizeof(uint32_t));
	(void)hipMalloc(&seqA.idx, (sz.num_indices + 1) * sizeof(uint16_t));
	(void)hipMalloc(&seqA.meshes, (sz.num_meshes + 1) * sizeof(vgx_mesh));
	CHECK(vgx_tessellate_emit(ctx, ps, devDraws, o.num_draws, &seqA, nullptr));
	// sequence B: the user meshes, uploaded as they came out of the decoder
	vgx_cache_desc a = {}, b = {};
	a.pos = seqA.pos; a.color = seqA.color; a.idx = seqA.idx; a.meshes = seqA.meshes; a.num_meshes = sz.num_meshes; a.num_vertices = sz.num_vertices; a.num_indices = sz.num_indices;
	float* bPos = nullptr; uint32_t* bCol = nullptr; uint32_t* bUV = nullptr; uint16_t* bIdx = nullptr; vgx_mesh* bMesh = nullptr;
	(void)hipMalloc(&bPos, triPos.size() * 4); (void)hipMalloc(&bCol, triCol.size() * 4); (void)hipMalloc(&bUV, triUV.size() * 4);
	(void)hipMalloc(&bIdx, triIdx.size() * 2); (void)hipMalloc(&bMesh, triMesh.size() * sizeof(vgx_mesh));
	(void)hipMemcpy(bPos, triPos.data(), triPos.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bCol, triCol.data(), triCol.size() * 4, hipMemcpyHostToDevice);
	(void)hipMemcpy(bUV, triUV.data(), triUV.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bIdx, triIdx.data(), triIdx.size() * 2, hipMemcpyHostToDevice);
	(void)hipMemcpy(bMesh, triMesh.data(), triMesh.size() * sizeof(vgx_mesh), hipMemcpyHostToDevice);
	b.pos = bPos; b.color = bCol; b.idx = bIdx; b.meshes = bMesh; b.num_meshes = o.num_tri_meshes; b.num_vertices = o.num_tri_vertices; b.num_indices = o.num_tri_indices;
	// the frame: both sequences interleaved by draw, assembled
	sz.num_vertices += o.num_tri_vertices; sz.num_indices += o.num_tri_indices; sz.num_meshes += o.num_tri_meshes;
	vgx_mesh_out out = {};
	out.cap_vertices = sz.num_vertices; out.cap_indices = sz.num_indices; out.cap_meshes = sz.num_meshes;
	(void)hipMalloc(&out.pos, sz.num_vertices * 2 * sizeof(float));
	(void)hipMalloc(&out.color, sz.num_vertices * sizeof(uint32_t));
	(void)hipMalloc(&out.idx, sz.num_indices * sizeof(uint16_t));
	(void)hipMalloc(&out.meshes, sz.num_meshes * sizeof(vgx_mesh));
	vgx_assembly as = {};
	as.cap_drawcmds = 2 * sz.num_vertices / maxVB + 2 + o.num_draws; // + one per possible state change
	(void)hipMalloc(&as.drawcmds, as.cap_drawcmds * sizeof(vgx_drawcmd));
	(void)hipMalloc(&as.dev_num_drawcmds, sizeof(uint64_t));
	as.max_vb_vertices = maxVB; as.flags = VGX_ASM_SPLIT_STATE;
	(void)hipMalloc(&as.uv, sz.num_vertices * 4); as.uv_bytes = 4; as.uv_value[0] = 0x003F003Fu; // the white pixel of the font atlas, int16 x 2
	CHECK(vgx_set_assembly(ctx, &as));
	vgx_sizes* devSizes = nullptr; uint32_t* devStatus = nullptr;
	(void)hipMalloc(&devSizes, sizeof(vgx_sizes)); (void)hipMalloc(&devStatus, sizeof(uint32_t));
	CHECK(vgx_merge_uv(ctx, &a, &b, nullptr, bUV, devDraws, o.num_draws, &out, devSizes, devStatus, nullptr)); // asynchronous: no host round trip inside
	(void)hipDeviceSynchronize();
	uint32_t status = 1; uint64_t ncmd = 0;
	(void)hipMemcpy(&status, devStatus, 4, hipMemcpyDeviceToHost);
	(void)hipMemcpy(&ncmd, as.dev_num_drawcmds, 8, hipMemcpyDeviceToHost);
	if (status != VGX_OK) { fprintf(stderr, "device status %u\n", status); return 1; }
	std::vector<vgx_drawcmd> cmds(ncmd);
	(void)hipMemcpy(cmds.data(), as.drawcmds, ncmd * sizeof(vgx_drawcmd), hipMemcpyDeviceToHost);

	// ---- what a bgfx back end would now do: one bgfx::update per vertex buffer stream, one for the index buffer, one submit per command ----
	uint64_t nv = 0, ni = 0; uint32_t nvb = 0; bool ok = true;
	for (uint64_t c = 0; c < ncmd; ++c) {
		const vgx_drawcmd& d = cmds[c];
		ok = ok && d.first_index == ni && d.num_vertices <= maxVB && d.first_vertex_in_vb + d.num_vertices <= maxVB;
		nv += d.num_vertices; ni += d.num_indices;
		if (d.vertex_buffer + 1 > nvb) { nvb = d.vertex_buffer + 1; }
	}
	ok = ok && nv == sz.num_vertices && ni == sz.num_indices;
	printf("frame: %llu vertices, %llu indices, %llu meshes -> %llu draw commands in %u vertex buffers of <= %u vertices: %s\n",
		(unsigned long long)sz.num_vertices, (unsigned long long)sz.num_indices, (unsigned long long)sz.num_meshes, (unsigned long long)ncmd, nvb, maxVB,
		ok ? "consistent" : "INCONSISTENT");
	printf("command 0: vertex buffer %u, first vertex %u, %u vertices, first index %llu, %u indices, state key 0x%x, scissor of its first draw %u %u %u %u\n",
		cmds[0].vertex_buffer, cmds[0].first_vertex_in_vb, cmds[0].num_vertices, (unsigned long long)cmds[0].first_index, cmds[0].num_indices, cmds[0].state_key,
		dstate[0].scissor[0], dstate[0].scissor[1], dstate[0].scissor[2], dstate[0].scissor[3]);
	CHECK(vgx_set_assembly(ctx, nullptr));
	CHECK(vgx_pathset_destroy(ctx, ps));
	CHECK(vgx_destroy(ctx));
	return ok ? 0 : 1;
}
