// Recorder behind oracle/bgfx_stub/bgfx/bgfx.h (see that header). TEST INFRASTRUCTURE ONLY.
#include "bgfx/bgfx.h"
#include <stdlib.h>

namespace bgfx
{
Stub g_stub;

Stub::Stub() : numTextures(0), scissorCacheNext(0)
{
	memset(view, 0, sizeof(view));
	memset(proj, 0, sizeof(proj));
	resetDraw();
}

void Stub::resetDraw()
{
	cur.view = 0; cur.program = kInvalidHandle;
	for (int i = 0; i < 3; ++i) { cur.vb[i] = kInvalidHandle; cur.vbFirst[i] = 0; cur.vbNum[i] = 0; }
	cur.ib = kInvalidHandle; cur.ibFirst = 0; cur.ibNum = 0;
	cur.scissor[0] = cur.scissor[1] = cur.scissor[2] = cur.scissor[3] = 0;
	cur.scissorCacheID = UINT16_MAX;
	cur.state = 0; cur.stencil = 0; cur.texture = kInvalidHandle; cur.textureFlags = 0;
	cur.uniforms.clear();
}

void Stub::resetFrame()
{
	submits.clear();
	scissorCache.clear();
	scissorCacheNext = 0;
	resetDraw();
}

RendererType::Enum getRendererType() { return RendererType::Noop; }
const Caps* getCaps()
{
	static Caps caps = { false, { 16384 } };
	return &caps;
}

static Memory* newMem(const void* data, uint32_t size, ReleaseFn rel, void* ud, bool owned)
{
	Memory* m = new Memory;
	m->data = (uint8_t*)data; m->size = size; m->release = rel; m->userData = ud; m->owned = owned;
	return m;
}
static void doneMem(const Memory* cm)
{
	Memory* m = const_cast<Memory*>(cm);
	if (m->release) { m->release(m->data, m->userData); }
	if (m->owned) { ::free(m->data); }
	delete m;
}
const Memory* alloc(uint32_t size) { return newMem(::malloc(size ? size : 1), size, nullptr, nullptr, true); }
const Memory* copy(const void* data, uint32_t size)
{
	void* p = ::malloc(size ? size : 1);
	memcpy(p, data, size);
	return newMem(p, size, nullptr, nullptr, true);
}
const Memory* makeRef(const void* data, uint32_t size, ReleaseFn rel, void* ud) { return newMem(data, size, rel, ud, false); }

ShaderHandle createEmbeddedShader(const EmbeddedShader*, RendererType::Enum, const char* name)
{
	g_stub.shaderNames.push_back(name);
	return ShaderHandle{ (uint16_t)(g_stub.shaderNames.size() - 1) };
}
ProgramHandle createProgram(ShaderHandle vsh, ShaderHandle fsh, bool)
{
	g_stub.programs.push_back((uint32_t)vsh.idx | ((uint32_t)fsh.idx << 16));
	return ProgramHandle{ (uint16_t)(g_stub.programs.size() - 1) };
}
UniformHandle createUniform(const char* name, UniformType::Enum, uint16_t)
{
	g_stub.uniformNames.push_back(name);
	return UniformHandle{ (uint16_t)(g_stub.uniformNames.size() - 1) };
}
DynamicVertexBufferHandle createDynamicVertexBuffer(uint32_t, const VertexLayout& layout, uint16_t)
{
	g_stub.vbs.push_back(StubBuffer{ {}, layout.stride, true });
	return DynamicVertexBufferHandle{ (uint16_t)(g_stub.vbs.size() - 1) };
}
DynamicIndexBufferHandle createDynamicIndexBuffer(const Memory* mem, uint16_t)
{
	g_stub.ibs.push_back(StubBuffer{ {}, 2, true });
	DynamicIndexBufferHandle h{ (uint16_t)(g_stub.ibs.size() - 1) };
	if (mem) { update(h, 0, mem); }
	return h;
}
TextureHandle createTexture2D(uint16_t, uint16_t, bool, uint16_t, TextureFormat::Enum, uint64_t, const Memory* mem)
{
	if (mem) { doneMem(mem); }
	return TextureHandle{ g_stub.numTextures++ };
}

static void store(StubBuffer& b, uint32_t start, const Memory* mem)
{
	const size_t off = (size_t)start * b.stride;
	if (b.bytes.size() < off + mem->size) { b.bytes.resize(off + mem->size); }
	if (mem->size) { memcpy(b.bytes.data() + off, mem->data, mem->size); }
	if (start == 0) { b.bytes.resize(mem->size); } // a whole-buffer update defines this frame's content
	doneMem(mem);
}
void update(DynamicVertexBufferHandle h, uint32_t start, const Memory* mem) { store(g_stub.vbs[h.idx], start, mem); }
void update(DynamicIndexBufferHandle h, uint32_t start, const Memory* mem) { store(g_stub.ibs[h.idx], start, mem); }
void updateTexture2D(TextureHandle, uint16_t, uint8_t, uint16_t, uint16_t, uint16_t, uint16_t, const Memory* mem, uint16_t) { doneMem(mem); }

void destroy(DynamicIndexBufferHandle h) { g_stub.ibs[h.idx].live = false; g_stub.ibs[h.idx].bytes.clear(); }
void destroy(DynamicVertexBufferHandle h) { g_stub.vbs[h.idx].live = false; g_stub.vbs[h.idx].bytes.clear(); }
void destroy(ProgramHandle) {}
void destroy(TextureHandle) {}
void destroy(UniformHandle) {}

void setViewTransform(ViewId, const void* view, const void* proj)
{
	memcpy(g_stub.view, view, sizeof(g_stub.view));
	memcpy(g_stub.proj, proj, sizeof(g_stub.proj));
}
void setVertexBuffer(uint8_t stream, DynamicVertexBufferHandle h, uint32_t start, uint32_t num)
{
	g_stub.cur.vb[stream] = h.idx; g_stub.cur.vbFirst[stream] = start; g_stub.cur.vbNum[stream] = num;
}
void setIndexBuffer(DynamicIndexBufferHandle h, uint32_t first, uint32_t num)
{
	g_stub.cur.ib = h.idx; g_stub.cur.ibFirst = first; g_stub.cur.ibNum = num;
}
uint16_t setScissor(uint16_t x, uint16_t y, uint16_t w, uint16_t h)
{
	const uint16_t id = g_stub.scissorCacheNext++;
	g_stub.scissorCache.push_back(x); g_stub.scissorCache.push_back(y); g_stub.scissorCache.push_back(w); g_stub.scissorCache.push_back(h);
	g_stub.cur.scissor[0] = x; g_stub.cur.scissor[1] = y; g_stub.cur.scissor[2] = w; g_stub.cur.scissor[3] = h;
	g_stub.cur.scissorCacheID = id;
	return id;
}
void setScissor(uint16_t cache)
{
	g_stub.cur.scissorCacheID = cache;
	if (cache != UINT16_MAX && (size_t)cache * 4 + 3 < g_stub.scissorCache.size()) {
		for (int i = 0; i < 4; ++i) { g_stub.cur.scissor[i] = g_stub.scissorCache[(size_t)cache * 4 + i]; }
	}
}
void setState(uint64_t state, uint32_t) { g_stub.cur.state = state; }
void setStencil(uint32_t f, uint32_t) { g_stub.cur.stencil = f; }
void setTexture(uint8_t, UniformHandle, TextureHandle h, uint32_t flags) { g_stub.cur.texture = h.idx; g_stub.cur.textureFlags = flags; }
void setUniform(UniformHandle h, const void* value, uint16_t)
{
	StubUniformValue u; u.handle = h.idx;
	memset(u.v, 0, sizeof(u.v));
	// Mat3 = 9 floats, Vec4 = 4; the reference's arrays are at least that long (vg.cpp:84-96)
	const char* nm = g_stub.uniformNames[h.idx];
	const size_t n = (nm && strcmp(nm, "u_paintMat") == 0) ? 9 : 4;
	memcpy(u.v, value, sizeof(float) * n);
	g_stub.cur.uniforms.push_back(u);
}
void submit(ViewId id, ProgramHandle program, uint32_t, uint8_t)
{
	g_stub.cur.view = id; g_stub.cur.program = program.idx;
	g_stub.submits.push_back(g_stub.cur);
	g_stub.resetDraw(); // BGFX_DISCARD_ALL, the default of submit()
}
}
