// ref_vg_capi.cpp -- C wrapper that drives the REFERENCE'S OWN src/vg.cpp (oracle/_ref/libvgref_vg.so).
// TEST INFRASTRUCTURE ONLY: only tests/ may load this. It is the parity oracle of the rows that live in vg.cpp
// (SURVEY 8(f)-1 draw-command assembly, 8(f)-2 command-list byte-code, 8(f)-3 shape cache): the reference's Context
// compiled UNMODIFIED where it lies under /root/reference (this translation unit #includes src/vg.cpp so that the
// wrapper can read Context / CommandList / CommandListCache, which are private to that file), linked with the
// reference's path.cpp, stroker.cpp, vg_util.cpp, libtess2 and fontstash, over oracle/bx_shim and the recording
// bgfx stand-in oracle/bgfx_stub. No reference source is copied into this repository.
//
// What a test can do with it:
//   - play a frame through the reference's immediate API (vg::beginPath .. vg::fillPath / strokePath, state and
//     transform calls) or record it with the reference's vg::clXxx writers and submit it, then read back what
//     vg::end() hands to bgfx: the frame's vertex buffers (pos / uv / colour), its index buffer and its draw commands
//     (vg.cpp:1076-1288, 5207-5460)                                                          -> checks vgx_set_assembly
//   - read a recorded CommandList::m_CommandBuffer byte for byte (vg.cpp:5694-5723)          -> input of vgx_cmdlist_decode
//   - read a cacheable list's CommandListCache (local-space meshes, vg.cpp:5773-5841) and the frame a cached
//     re-submission produces (clCacheRender / submitCachedMesh, :5845-6211)                  -> checks vgx_cache_*
#include "vg.cpp" // /root/reference/src/vg.cpp (found through -I$(REF)/src)

#include <bx/allocator.h>

#if VGR_WITH_COMPAT
// oracle/_ref/libvgref_vg_compat.so: the same Context, but vg::createPath .. vg::strokerConcaveFillEndAA resolve to the PRODUCT's
// libvgx_compat.so instead of the reference's path.cpp / stroker.cpp (which are not linked); libtess2 is the caller's, as in an
// application: handed over once.
#include "vgx_compat.hpp"
#include "libtess2/tesselator.h"
static void vgrInstallTess()
{
	static const vg::VgxTessApi api = { (void* (*)(void*))tessNewTess, (void (*)(void*))tessDeleteTess, (void (*)(void*, int, const void*, int, int))tessAddContour,
		(int (*)(void*, int, int, int, int, const float*))tessTesselate, (int (*)(void*))tessGetVertexCount, (const float* (*)(void*))tessGetVertices,
		(int (*)(void*))tessGetElementCount, (const unsigned short* (*)(void*))tessGetElements };
	vg::vgxCompatSetTessellator(&api);
}
#endif

namespace {
struct Ref
{
	bx::ShimAllocator alloc;
	vg::Context* ctx;
};

inline vg::CommandListHandle clh(uint32_t h) { vg::CommandListHandle r = { (uint16_t)h }; return r; }
}

extern "C" {

struct vgr_drawcmd // vg::DrawCommand (vg.cpp:98-131), flattened
{
	uint32_t type;
	uint32_t vertex_buffer;
	uint32_t first_vertex;
	uint32_t first_index;
	uint32_t num_vertices;
	uint32_t num_indices;
	uint16_t scissor[4];
	uint32_t handle;
	uint32_t clip_rule;
	uint32_t clip_first_cmd;
	uint32_t clip_num_cmds;
};

struct vgr_submit // one bgfx::submit as vg::end() issued it (vg.cpp:1160-1288)
{
	uint32_t program;      // index into Context::m_ProgramHandle = DrawCommand::Type
	uint32_t vb_pos;       // frame-relative vertex buffer (stream 0)
	uint32_t first_vertex;
	uint32_t num_vertices;
	uint32_t first_index;
	uint32_t num_indices;
	uint32_t has_color_stream;
	uint32_t has_uv_stream;
	uint16_t scissor[4];
	uint32_t stencil;
	uint32_t texture;      // bgfx texture handle or 0xFFFF
	uint64_t state;
	uint32_t num_uniforms;
	uint32_t reserved;
	float paint_mat[9];    // u_paintMat if set (gradient / image pattern)
	float params[4];       // u_extentRadiusFeather
	float inner_color[4];
	float outer_color[4];
};

static int g_liveContexts = 0;
void* vgr_create(uint32_t maxVBVertices, uint32_t maxCommandLists, uint32_t maxGradients, uint32_t maxImagePatterns)
{
	Ref* r = new Ref;
	++g_liveContexts;
#if VGR_WITH_COMPAT
	vgrInstallTess();
#endif
	vg::ContextConfig cfg;
	cfg.m_MaxGradients = (uint16_t)(maxGradients ? maxGradients : 64);
	cfg.m_MaxImagePatterns = (uint16_t)(maxImagePatterns ? maxImagePatterns : 64);
	cfg.m_MaxFonts = 8;
	cfg.m_MaxStateStackSize = 32;
	cfg.m_MaxImages = 16;
	cfg.m_MaxCommandLists = (uint16_t)(maxCommandLists ? maxCommandLists : 256);
	cfg.m_MaxVBVertices = maxVBVertices ? maxVBVertices : 65536;
	cfg.m_FontAtlasImageFlags = vg::ImageFlags::Filter_Bilinear;
	cfg.m_MaxCommandListDepth = 16;
	cfg.m_ResetViewTransformOnEnd = true;
	r->ctx = vg::createContext(&r->alloc, &cfg);
	return r;
}

void vgr_destroy(void* h)
{
	Ref* r = (Ref*)h;
	vg::destroyContext(r->ctx);
	delete r;
	// The stand-in hands out buffer / texture / shader handles by counting up (uint16): with no Context alive nothing refers
	// to them any more, so start over -- a test process may create tens of thousands of Contexts one after the other.
	if (--g_liveContexts == 0) { bgfx::g_stub = bgfx::Stub(); }
}

void vgr_begin(void* h, uint32_t w, uint32_t hgt, float dpr)
{
	bgfx::g_stub.resetFrame();
	vg::begin(((Ref*)h)->ctx, 0, (uint16_t)w, (uint16_t)hgt, dpr);
}
void vgr_end(void* h) { vg::end(((Ref*)h)->ctx); }
void vgr_frame(void* h) { vg::frame(((Ref*)h)->ctx); }

uint32_t vgr_cl_create(void* h, uint32_t flags) { return vg::createCommandList(((Ref*)h)->ctx, flags).idx; }
void vgr_cl_destroy(void* h, uint32_t cl) { vg::destroyCommandList(((Ref*)h)->ctx, clh(cl)); }
void vgr_cl_reset(void* h, uint32_t cl) { vg::resetCommandList(((Ref*)h)->ctx, clh(cl)); }
uint32_t vgr_create_image(void* h, uint32_t w, uint32_t hgt, uint32_t flags) { return vg::createImage(((Ref*)h)->ctx, (uint16_t)w, (uint16_t)hgt, flags, nullptr).idx; }

// CommandList::m_CommandBuffer / m_CommandBufferPos and the local handle counts (vg.cpp:229-241)
int vgr_cl_bytes(void* h, uint32_t cl, const uint8_t** bytes, uint32_t* size, uint32_t* numGradients, uint32_t* numImagePatterns)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	if (!vg::isCommandListHandleValid(ctx, clh(cl))) { return 1; }
	const vg::CommandList* c = &ctx->m_CmdLists[cl];
	*bytes = c->m_CommandBuffer; *size = c->m_CommandBufferPos;
	if (numGradients) { *numGradients = c->m_NumGradients; }
	if (numImagePatterns) { *numImagePatterns = c->m_NumImagePatterns; }
	return 0;
}

// One entry point for every vg::xxx / vg::clXxx call a test needs. `op` = vg::CommandType::Enum (vg.cpp:177-241);
// cl == 0xFFFFFFFF plays the call on the Context (immediate mode), otherwise it is recorded into that command list
// with the reference's own writer. f = float arguments in the reference's parameter order, u = integer arguments
// (documented per case). Returns the handle of the Create* calls (idx | flags << 16), else 0.
uint32_t vgr_op(void* h, uint32_t cl, uint32_t op, const float* f, const uint32_t* u)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	const bool rec = cl != 0xFFFFFFFFu;
	const vg::CommandListHandle L = clh(cl);
	using CT = vg::CommandType;
	switch (op) {
	case CT::BeginPath: rec ? vg::clBeginPath(ctx, L) : vg::beginPath(ctx); break;
	case CT::MoveTo: rec ? vg::clMoveTo(ctx, L, f[0], f[1]) : vg::moveTo(ctx, f[0], f[1]); break;
	case CT::LineTo: rec ? vg::clLineTo(ctx, L, f[0], f[1]) : vg::lineTo(ctx, f[0], f[1]); break;
	case CT::CubicTo: rec ? vg::clCubicTo(ctx, L, f[0], f[1], f[2], f[3], f[4], f[5]) : vg::cubicTo(ctx, f[0], f[1], f[2], f[3], f[4], f[5]); break;
	case CT::QuadraticTo: rec ? vg::clQuadraticTo(ctx, L, f[0], f[1], f[2], f[3]) : vg::quadraticTo(ctx, f[0], f[1], f[2], f[3]); break;
	case CT::ArcTo: rec ? vg::clArcTo(ctx, L, f[0], f[1], f[2], f[3], f[4]) : vg::arcTo(ctx, f[0], f[1], f[2], f[3], f[4]); break;
	case CT::Arc: rec ? vg::clArc(ctx, L, f[0], f[1], f[2], f[3], f[4], (vg::Winding::Enum)u[0]) : vg::arc(ctx, f[0], f[1], f[2], f[3], f[4], (vg::Winding::Enum)u[0]); break;
	case CT::Rect: rec ? vg::clRect(ctx, L, f[0], f[1], f[2], f[3]) : vg::rect(ctx, f[0], f[1], f[2], f[3]); break;
	case CT::RoundedRect: rec ? vg::clRoundedRect(ctx, L, f[0], f[1], f[2], f[3], f[4]) : vg::roundedRect(ctx, f[0], f[1], f[2], f[3], f[4]); break;
	case CT::RoundedRectVarying: rec ? vg::clRoundedRectVarying(ctx, L, f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]) : vg::roundedRectVarying(ctx, f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]); break;
	case CT::Circle: rec ? vg::clCircle(ctx, L, f[0], f[1], f[2]) : vg::circle(ctx, f[0], f[1], f[2]); break;
	case CT::Ellipse: rec ? vg::clEllipse(ctx, L, f[0], f[1], f[2], f[3]) : vg::ellipse(ctx, f[0], f[1], f[2], f[3]); break;
	case CT::Polyline: rec ? vg::clPolyline(ctx, L, f, u[0]) : vg::polyline(ctx, f, u[0]); break; // u[0] = numPoints
	case CT::ClosePath: rec ? vg::clClosePath(ctx, L) : vg::closePath(ctx); break;
	case CT::FillPathColor: rec ? vg::clFillPath(ctx, L, (vg::Color)u[0], u[1]) : vg::fillPath(ctx, (vg::Color)u[0], u[1]); break; // u = {color, flags}
	case CT::FillPathGradient: { // u = {handle idx, flags, handle flags}
		const vg::GradientHandle g = { (uint16_t)u[0], (uint16_t)u[2] };
		rec ? vg::clFillPath(ctx, L, g, u[1]) : vg::fillPath(ctx, g, u[1]);
	} break;
	case CT::FillPathImagePattern: { // u = {handle idx, flags, handle flags, color}
		const vg::ImagePatternHandle g = { (uint16_t)u[0], (uint16_t)u[2] };
		rec ? vg::clFillPath(ctx, L, g, (vg::Color)u[3], u[1]) : vg::fillPath(ctx, g, (vg::Color)u[3], u[1]);
	} break;
	case CT::StrokePathColor: rec ? vg::clStrokePath(ctx, L, (vg::Color)u[0], f[0], u[1]) : vg::strokePath(ctx, (vg::Color)u[0], f[0], u[1]); break; // f = {width}, u = {color, flags}
	case CT::StrokePathGradient: {
		const vg::GradientHandle g = { (uint16_t)u[0], (uint16_t)u[2] };
		rec ? vg::clStrokePath(ctx, L, g, f[0], u[1]) : vg::strokePath(ctx, g, f[0], u[1]);
	} break;
	case CT::StrokePathImagePattern: {
		const vg::ImagePatternHandle g = { (uint16_t)u[0], (uint16_t)u[2] };
		rec ? vg::clStrokePath(ctx, L, g, (vg::Color)u[3], f[0], u[1]) : vg::strokePath(ctx, g, (vg::Color)u[3], f[0], u[1]);
	} break;
	case CT::BeginClip: rec ? vg::clBeginClip(ctx, L, (vg::ClipRule::Enum)u[0]) : vg::beginClip(ctx, (vg::ClipRule::Enum)u[0]); break;
	case CT::EndClip: rec ? vg::clEndClip(ctx, L) : vg::endClip(ctx); break;
	case CT::ResetClip: rec ? vg::clResetClip(ctx, L) : vg::resetClip(ctx); break;
	case CT::CreateLinearGradient: { // f = {sx, sy, ex, ey}, u = {icol, ocol}
		const vg::GradientHandle g = rec ? vg::clCreateLinearGradient(ctx, L, f[0], f[1], f[2], f[3], u[0], u[1]) : vg::createLinearGradient(ctx, f[0], f[1], f[2], f[3], u[0], u[1]);
		return (uint32_t)g.idx | ((uint32_t)g.flags << 16);
	}
	case CT::CreateBoxGradient: {
		const vg::GradientHandle g = rec ? vg::clCreateBoxGradient(ctx, L, f[0], f[1], f[2], f[3], f[4], f[5], u[0], u[1]) : vg::createBoxGradient(ctx, f[0], f[1], f[2], f[3], f[4], f[5], u[0], u[1]);
		return (uint32_t)g.idx | ((uint32_t)g.flags << 16);
	}
	case CT::CreateRadialGradient: {
		const vg::GradientHandle g = rec ? vg::clCreateRadialGradient(ctx, L, f[0], f[1], f[2], f[3], u[0], u[1]) : vg::createRadialGradient(ctx, f[0], f[1], f[2], f[3], u[0], u[1]);
		return (uint32_t)g.idx | ((uint32_t)g.flags << 16);
	}
	case CT::CreateImagePattern: { // f = {cx, cy, w, h, angle}, u = {image}
		const vg::ImageHandle img = { (uint16_t)u[0] };
		const vg::ImagePatternHandle g = rec ? vg::clCreateImagePattern(ctx, L, f[0], f[1], f[2], f[3], f[4], img) : vg::createImagePattern(ctx, f[0], f[1], f[2], f[3], f[4], img);
		return (uint32_t)g.idx | ((uint32_t)g.flags << 16);
	}
	case CT::PushState: rec ? vg::clPushState(ctx, L) : vg::pushState(ctx); break;
	case CT::PopState: rec ? vg::clPopState(ctx, L) : vg::popState(ctx); break;
	case CT::ResetScissor: rec ? vg::clResetScissor(ctx, L) : vg::resetScissor(ctx); break;
	case CT::SetScissor: rec ? vg::clSetScissor(ctx, L, f[0], f[1], f[2], f[3]) : vg::setScissor(ctx, f[0], f[1], f[2], f[3]); break;
	case CT::IntersectScissor: if (rec) { vg::clIntersectScissor(ctx, L, f[0], f[1], f[2], f[3]); } else { vg::intersectScissor(ctx, f[0], f[1], f[2], f[3]); } break;
	case CT::TransformIdentity: rec ? vg::clTransformIdentity(ctx, L) : vg::transformIdentity(ctx); break;
	case CT::TransformScale: rec ? vg::clTransformScale(ctx, L, f[0], f[1]) : vg::transformScale(ctx, f[0], f[1]); break;
	case CT::TransformTranslate: rec ? vg::clTransformTranslate(ctx, L, f[0], f[1]) : vg::transformTranslate(ctx, f[0], f[1]); break;
	case CT::TransformRotate: rec ? vg::clTransformRotate(ctx, L, f[0]) : vg::transformRotate(ctx, f[0]); break;
	case CT::TransformMult: rec ? vg::clTransformMult(ctx, L, f, (vg::TransformOrder::Enum)u[0]) : vg::transformMult(ctx, f, (vg::TransformOrder::Enum)u[0]); break;
	case CT::SetViewBox: rec ? vg::clSetViewBox(ctx, L, f[0], f[1], f[2], f[3]) : vg::setViewBox(ctx, f[0], f[1], f[2], f[3]); break;
	case CT::SetGlobalAlpha: rec ? vg::clSetGlobalAlpha(ctx, L, f[0]) : vg::setGlobalAlpha(ctx, f[0]); break;
	case CT::IndexedTriList: { // f = pos[2 * nv]; u = {nv, hasUV, nc, ni, image idx (0xFFFF = invalid), colours[nc], uv words[hasUV ? nv * sizeof(uv_t) * 2 / 4 : 0], indices packed two per word}
		const uint32_t nv = u[0], hasUV = u[1], nc = u[2], ni = u[3];
		const vg::ImageHandle img = { (uint16_t)u[4] };
		const vg::Color* col = (const vg::Color*)(u + 5);
		const uint32_t uvWords = hasUV ? nv * (uint32_t)(sizeof(vg::uv_t) * 2 / 4) : 0;
		const vg::uv_t* uv = hasUV ? (const vg::uv_t*)(u + 5 + nc) : nullptr;
		const uint16_t* idx = (const uint16_t*)(u + 5 + nc + uvWords);
		if (rec) { vg::clIndexedTriList(ctx, L, f, uv, nv, col, nc, idx, ni, img); } else { vg::indexedTriList(ctx, f, uv, nv, col, nc, idx, ni, img); }
	} break;
	case CT::SubmitCommandList: rec ? vg::clSubmitCommandList(ctx, L, clh(u[0])) : vg::submitCommandList(ctx, clh(u[0])); break; // u = {child}
	default: return 0xFFFFFFFFu;
	}
	return 0;
}

// State on top of the stack (vg.cpp:62-69)
void vgr_get_state(void* h, float* mtx6, float* scissor4, float* alphaAvgFont3)
{
	const vg::State* s = vg::getState(((Ref*)h)->ctx);
	memcpy(mtx6, s->m_TransformMtx, sizeof(float) * 6);
	memcpy(scissor4, s->m_ScissorRect, sizeof(float) * 4);
	alphaAvgFont3[0] = s->m_GlobalAlpha; alphaAvgFont3[1] = s->m_AvgScale; alphaAvgFont3[2] = s->m_FontScale;
}
void vgr_get_params(void* h, float* tessTolFringe2)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	tessTolFringe2[0] = ctx->m_TesselationTolerance; tessTolFringe2[1] = ctx->m_FringeWidth;
}
uint32_t vgr_font_image(void* h) { return ((Ref*)h)->ctx->m_FontImages[0].idx; } // the image ctxIndexedTriList falls back to (vg.cpp:4131-4133)
void vgr_white_uv(void* h, void* out, uint32_t* bytesPerUV)
{
	const vg::uv_t* uv = vg::getWhitePixelUV(((Ref*)h)->ctx);
	memcpy(out, uv, sizeof(vg::uv_t) * 2);
	*bytesPerUV = (uint32_t)(sizeof(vg::uv_t) * 2);
}

// ---- the frame vg::end() handed to bgfx ------------------------------------------------------------------------------
uint32_t vgr_num_vertex_buffers(void* h)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	return ctx->m_NumVertexBuffers - ctx->m_FirstVertexBufferID;
}
// stream: 0 = pos (float x 2), 1 = uv (uv_t x 2), 2 = colour (uint32)
int vgr_vertex_buffer(void* h, uint32_t i, uint32_t stream, const void** bytes, uint64_t* size)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	const uint32_t id = ctx->m_FirstVertexBufferID + i;
	if (id >= ctx->m_NumVertexBuffers) { return 1; }
	const vg::GPUVertexBuffer* g = &ctx->m_GPUVertexBuffers[id];
	const uint16_t hd = stream == 0 ? g->m_PosBufferHandle.idx : (stream == 1 ? g->m_UVBufferHandle.idx : g->m_ColorBufferHandle.idx);
	if (hd == bgfx::kInvalidHandle) { return 2; }
	const bgfx::StubBuffer& b = bgfx::g_stub.vbs[hd];
	*bytes = b.bytes.data(); *size = b.bytes.size();
	return 0;
}
int vgr_index_buffer(void* h, const void** bytes, uint64_t* size)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	const uint16_t hd = ctx->m_GPUIndexBuffers[ctx->m_ActiveIndexBufferID].m_bgfxHandle.idx;
	if (hd == bgfx::kInvalidHandle) { return 2; }
	const bgfx::StubBuffer& b = bgfx::g_stub.ibs[hd];
	*bytes = b.bytes.data(); *size = b.bytes.size();
	return 0;
}
static void fillCmd(const vg::Context* ctx, const vg::DrawCommand* c, vgr_drawcmd* o)
{
	o->type = (uint32_t)c->m_Type;
	o->vertex_buffer = c->m_VertexBufferID - ctx->m_FirstVertexBufferID;
	o->first_vertex = c->m_FirstVertexID; o->first_index = c->m_FirstIndexID;
	o->num_vertices = c->m_NumVertices; o->num_indices = c->m_NumIndices;
	memcpy(o->scissor, c->m_ScissorRect, sizeof(o->scissor));
	o->handle = c->m_HandleID;
	o->clip_rule = (uint32_t)c->m_ClipState.m_Rule; o->clip_first_cmd = c->m_ClipState.m_FirstCmdID; o->clip_num_cmds = c->m_ClipState.m_NumCmds;
}
uint32_t vgr_draw_commands(void* h, vgr_drawcmd* out, uint32_t cap)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	for (uint32_t i = 0; i < ctx->m_NumDrawCommands && i < cap; ++i) { fillCmd(ctx, &ctx->m_DrawCommands[i], &out[i]); }
	return ctx->m_NumDrawCommands;
}
uint32_t vgr_clip_commands(void* h, vgr_drawcmd* out, uint32_t cap)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	for (uint32_t i = 0; i < ctx->m_NumClipCommands && i < cap; ++i) { fillCmd(ctx, &ctx->m_ClipCommands[i], &out[i]); }
	return ctx->m_NumClipCommands;
}
uint32_t vgr_submits(void* h, vgr_submit* out, uint32_t cap)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	const std::vector<bgfx::StubSubmit>& S = bgfx::g_stub.submits;
	for (uint32_t i = 0; i < S.size() && i < cap; ++i) {
		const bgfx::StubSubmit& s = S[i];
		vgr_submit* o = &out[i];
		memset(o, 0, sizeof(*o));
		o->program = 0xFFFFFFFFu;
		for (uint32_t t = 0; t < vg::DrawCommand::Type::NumTypes; ++t) { if (ctx->m_ProgramHandle[t].idx == s.program) { o->program = t; } }
		o->vb_pos = 0xFFFFFFFFu;
		for (uint32_t v = ctx->m_FirstVertexBufferID; v < ctx->m_NumVertexBuffers; ++v) { if (ctx->m_GPUVertexBuffers[v].m_PosBufferHandle.idx == s.vb[0]) { o->vb_pos = v - ctx->m_FirstVertexBufferID; } }
		o->first_vertex = s.vbFirst[0]; o->num_vertices = s.vbNum[0];
		o->first_index = s.ibFirst; o->num_indices = s.ibNum;
		o->has_color_stream = s.vb[1] != bgfx::kInvalidHandle; o->has_uv_stream = s.vb[2] != bgfx::kInvalidHandle;
		memcpy(o->scissor, s.scissor, sizeof(o->scissor));
		o->stencil = s.stencil; o->texture = s.texture; o->state = s.state;
		o->num_uniforms = (uint32_t)s.uniforms.size();
		for (const bgfx::StubUniformValue& uv : s.uniforms) {
			if (uv.handle == ctx->m_PaintMatUniform.idx) { memcpy(o->paint_mat, uv.v, sizeof(float) * 9); }
			else if (uv.handle == ctx->m_ExtentRadiusFeatherUniform.idx) { memcpy(o->params, uv.v, sizeof(float) * 4); }
			else if (uv.handle == ctx->m_InnerColorUniform.idx) { memcpy(o->inner_color, uv.v, sizeof(float) * 4); }
			else if (uv.handle == ctx->m_OuterColorUniform.idx) { memcpy(o->outer_color, uv.v, sizeof(float) * 4); }
		}
	}
	return (uint32_t)S.size();
}

// ---- shape cache of a Cacheable command list (vg.cpp:249-275) --------------------------------------------------------
int vgr_cache_info(void* h, uint32_t cl, uint32_t* numMeshes, uint32_t* numCommands, float* avgScale)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	if (!vg::isCommandListHandleValid(ctx, clh(cl))) { return 1; }
	const vg::CommandListCache* c = ctx->m_CmdLists[cl].m_Cache;
	if (!c) { return 2; }
	*numMeshes = c->m_NumMeshes; *numCommands = c->m_NumCommands; *avgScale = c->m_AvgScale;
	return 0;
}
int vgr_cache_mesh(void* h, uint32_t cl, uint32_t i, const float** pos, const uint32_t** colors, const uint16_t** idx, uint32_t* nv, uint32_t* ni)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	const vg::CommandListCache* c = ctx->m_CmdLists[cl].m_Cache;
	if (!c || i >= c->m_NumMeshes) { return 1; }
	const vg::CachedMesh* m = &c->m_Meshes[i];
	*pos = m->m_Pos; *colors = m->m_Colors; *idx = m->m_Indices; *nv = m->m_NumVertices; *ni = m->m_NumIndices;
	return 0;
}
int vgr_cache_command(void* h, uint32_t cl, uint32_t i, uint32_t* firstMesh, uint32_t* numMeshes, float* inv6)
{
	vg::Context* ctx = ((Ref*)h)->ctx;
	const vg::CommandListCache* c = ctx->m_CmdLists[cl].m_Cache;
	if (!c || i >= c->m_NumCommands) { return 1; }
	*firstMesh = c->m_Commands[i].m_FirstMeshID; *numMeshes = c->m_Commands[i].m_NumMeshes;
	memcpy(inv6, c->m_Commands[i].m_InvTransformMtx, sizeof(float) * 6);
	return 0;
}

const char* vgr_engine_name(void) { return "reference(vg-renderer src/vg.cpp @ /root/reference, scalar build, bx_shim + bgfx_stub)"; }

} // extern "C"
