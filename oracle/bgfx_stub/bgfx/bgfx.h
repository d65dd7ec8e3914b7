// Stand-in for <bgfx/bgfx.h>: just enough of the bgfx C++ API for the UNMODIFIED reference src/vg.cpp to compile
// and run without a renderer (oracle/_ref/libvgref_vg.so). Nothing is drawn: the calls vg::end() makes to hand a
// frame to bgfx (vg.cpp:1076-1288: makeRef / update / createDynamicIndexBuffer, setVertexBuffer / setIndexBuffer /
// setScissor / setStencil / setTexture / setUniform / submit) are RECORDED in bgfx::g_stub so that a test can read
// back exactly the vertex / index buffers and the draw calls the reference produces for a frame.
// TEST INFRASTRUCTURE ONLY (parity oracle of SURVEY 8(f)-1..3). bgfx is an un-vendored dependency of the reference.
#ifndef BGFX_STUB_BGFX_H
#define BGFX_STUB_BGFX_H

#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <vector>

#define BGFX_INVALID_HANDLE { bgfx::kInvalidHandle }

#define BGFX_BUFFER_NONE UINT16_C(0x0000)
#define BGFX_BUFFER_ALLOW_RESIZE UINT16_C(0x0800)

// sampler / texture flags: distinct bits, only compared and or-ed by vg.cpp (:2190-2230)
#define BGFX_TEXTURE_NONE UINT64_C(0x0000000000000000)
#define BGFX_SAMPLER_NONE UINT32_C(0x00000000)
#define BGFX_SAMPLER_U_CLAMP UINT32_C(0x00000002)
#define BGFX_SAMPLER_V_CLAMP UINT32_C(0x00000008)
#define BGFX_SAMPLER_W_CLAMP UINT32_C(0x00000020)
#define BGFX_SAMPLER_MIN_POINT UINT32_C(0x00000040)
#define BGFX_SAMPLER_MAG_POINT UINT32_C(0x00000100)
#define BGFX_SAMPLER_MIP_POINT UINT32_C(0x00000400)

// render state: opaque bit patterns, recorded as submitted
#define BGFX_STATE_WRITE_R UINT64_C(0x0000000000000001)
#define BGFX_STATE_WRITE_G UINT64_C(0x0000000000000002)
#define BGFX_STATE_WRITE_B UINT64_C(0x0000000000000004)
#define BGFX_STATE_WRITE_A UINT64_C(0x0000000000000008)
#define BGFX_STATE_WRITE_RGB (BGFX_STATE_WRITE_R | BGFX_STATE_WRITE_G | BGFX_STATE_WRITE_B)
#define BGFX_STATE_BLEND_ONE UINT64_C(0x0000000000002000)
#define BGFX_STATE_BLEND_SRC_ALPHA UINT64_C(0x0000000000005000)
#define BGFX_STATE_BLEND_INV_SRC_ALPHA UINT64_C(0x0000000000006000)
#define BGFX_STATE_BLEND_FUNC_SEPARATE(_srcRGB, _dstRGB, _srcA, _dstA) \
	(UINT64_C(0) | (((uint64_t)(_srcRGB) | ((uint64_t)(_dstRGB) << 4))) | (((uint64_t)(_srcA) | ((uint64_t)(_dstA) << 4)) << 8))

#define BGFX_STENCIL_NONE UINT32_C(0x00000000)
#define BGFX_STENCIL_FUNC_REF(v) (((uint32_t)(v)) & UINT32_C(0x000000ff))
#define BGFX_STENCIL_FUNC_RMASK(v) ((((uint32_t)(v)) << 8) & UINT32_C(0x0000ff00))
#define BGFX_STENCIL_TEST_EQUAL UINT32_C(0x00030000)
#define BGFX_STENCIL_TEST_NOTEQUAL UINT32_C(0x00060000)
#define BGFX_STENCIL_TEST_ALWAYS UINT32_C(0x00080000)
#define BGFX_STENCIL_OP_FAIL_S_KEEP UINT32_C(0x00100000)
#define BGFX_STENCIL_OP_FAIL_S_REPLACE UINT32_C(0x00200000)
#define BGFX_STENCIL_OP_FAIL_Z_KEEP UINT32_C(0x01000000)
#define BGFX_STENCIL_OP_FAIL_Z_REPLACE UINT32_C(0x02000000)
#define BGFX_STENCIL_OP_PASS_Z_KEEP UINT32_C(0x10000000)
#define BGFX_STENCIL_OP_PASS_Z_REPLACE UINT32_C(0x20000000)

namespace bgfx
{
static const uint16_t kInvalidHandle = UINT16_MAX;

#define BGFX_STUB_HANDLE(_name) \
	struct _name { uint16_t idx; }; \
	inline bool isValid(_name h) { return h.idx != kInvalidHandle; }
BGFX_STUB_HANDLE(DynamicIndexBufferHandle)
BGFX_STUB_HANDLE(DynamicVertexBufferHandle)
BGFX_STUB_HANDLE(ProgramHandle)
BGFX_STUB_HANDLE(ShaderHandle)
BGFX_STUB_HANDLE(TextureHandle)
BGFX_STUB_HANDLE(UniformHandle)
#undef BGFX_STUB_HANDLE

typedef uint16_t ViewId;
typedef void (*ReleaseFn)(void* ptr, void* userData);

struct Memory
{
	uint8_t* data;
	uint32_t size;
	// stub bookkeeping (real bgfx keeps these in a private subclass)
	ReleaseFn release;
	void* userData;
	bool owned;
};

struct RendererType { enum Enum { Noop, Count }; };
struct Attrib { enum Enum { Position, Normal, Tangent, Bitangent, Color0, Color1, Color2, Color3, Indices, Weight, TexCoord0, TexCoord1, Count }; };
struct AttribType { enum Enum { Uint8, Uint10, Int16, Half, Float, Count }; };
struct UniformType { enum Enum { Sampler, End, Vec4, Mat3, Mat4, Count }; };
struct TextureFormat { enum Enum { RGBA8, Count }; };

struct Caps
{
	bool homogeneousDepth;
	struct Limits { uint32_t maxTextureSize; } limits;
};

struct VertexLayout
{
	uint16_t stride;
	uint8_t numAttribs;
	struct A { uint8_t attrib, num, type, normalized; } attribs[4];
	VertexLayout& begin() { stride = 0; numAttribs = 0; return *this; }
	VertexLayout& add(Attrib::Enum a, uint8_t num, AttribType::Enum t, bool normalized = false, bool asInt = false)
	{
		(void)asInt;
		static const uint8_t sz[] = { 1, 4, 2, 2, 4 };
		attribs[numAttribs++] = A{ (uint8_t)a, num, (uint8_t)t, (uint8_t)normalized };
		stride = (uint16_t)(stride + (t == AttribType::Uint10 ? 4 : sz[t] * num));
		return *this;
	}
	void end() {}
};

struct EmbeddedShader { const char* name; };
#define BGFX_EMBEDDED_SHADER(_name) { #_name }
#define BGFX_EMBEDDED_SHADER_END() { nullptr }

// ---- the recorder ----------------------------------------------------------------------------------------------
struct StubBuffer { std::vector<uint8_t> bytes; uint16_t stride; bool live; };
struct StubUniformValue { uint16_t handle; float v[16]; };
struct StubSubmit
{
	uint16_t view;
	uint16_t program;
	uint16_t vb[3];          // stream 0..2 handle (kInvalidHandle when not bound)
	uint32_t vbFirst[3];
	uint32_t vbNum[3];
	uint16_t ib;
	uint32_t ibFirst;
	uint32_t ibNum;
	uint16_t scissor[4];     // x, y, w, h as last passed to setScissor(x, y, w, h)
	uint16_t scissorCacheID; // what setScissor(cache) selected; UINT16_MAX = none
	uint64_t state;
	uint32_t stencil;
	uint16_t texture;        // kInvalidHandle when none
	uint32_t textureFlags;
	std::vector<StubUniformValue> uniforms;
};

struct Stub
{
	std::vector<StubBuffer> vbs;
	std::vector<StubBuffer> ibs;
	std::vector<StubSubmit> submits;
	std::vector<Memory*> memPool;
	std::vector<const char*> uniformNames;
	std::vector<const char*> shaderNames;
	std::vector<uint32_t> programs; // vs | fs << 16
	uint16_t numTextures;
	uint16_t scissorCacheNext;
	std::vector<uint16_t> scissorCache; // 4 per entry
	StubSubmit cur;
	float view[16], proj[16];
	Stub();
	void resetDraw();
	void resetFrame(); // forget the submits (buffers persist like GPU buffers do)
};
extern Stub g_stub;

RendererType::Enum getRendererType();
const Caps* getCaps();

const Memory* alloc(uint32_t size);
const Memory* copy(const void* data, uint32_t size);
const Memory* makeRef(const void* data, uint32_t size, ReleaseFn releaseFn = nullptr, void* userData = nullptr);

ShaderHandle createEmbeddedShader(const EmbeddedShader* es, RendererType::Enum type, const char* name);
ProgramHandle createProgram(ShaderHandle vsh, ShaderHandle fsh, bool destroyShaders = false);
UniformHandle createUniform(const char* name, UniformType::Enum type, uint16_t num = 1);
DynamicVertexBufferHandle createDynamicVertexBuffer(uint32_t num, const VertexLayout& layout, uint16_t flags = BGFX_BUFFER_NONE);
DynamicIndexBufferHandle createDynamicIndexBuffer(const Memory* mem, uint16_t flags = BGFX_BUFFER_NONE);
TextureHandle createTexture2D(uint16_t width, uint16_t height, bool hasMips, uint16_t numLayers, TextureFormat::Enum format, uint64_t flags = 0, const Memory* mem = nullptr);

void update(DynamicVertexBufferHandle h, uint32_t startVertex, const Memory* mem);
void update(DynamicIndexBufferHandle h, uint32_t startIndex, const Memory* mem);
void updateTexture2D(TextureHandle h, uint16_t layer, uint8_t mip, uint16_t x, uint16_t y, uint16_t w, uint16_t hgt, const Memory* mem, uint16_t pitch = UINT16_MAX);

void destroy(DynamicIndexBufferHandle h);
void destroy(DynamicVertexBufferHandle h);
void destroy(ProgramHandle h);
void destroy(TextureHandle h);
void destroy(UniformHandle h);

void setViewTransform(ViewId id, const void* view, const void* proj);
void setVertexBuffer(uint8_t stream, DynamicVertexBufferHandle h, uint32_t startVertex, uint32_t numVertices);
void setIndexBuffer(DynamicIndexBufferHandle h, uint32_t firstIndex, uint32_t numIndices);
uint16_t setScissor(uint16_t x, uint16_t y, uint16_t w, uint16_t h);
void setScissor(uint16_t cache = UINT16_MAX);
void setState(uint64_t state, uint32_t rgba = 0);
void setStencil(uint32_t fstencil, uint32_t bstencil = BGFX_STENCIL_NONE);
void setTexture(uint8_t stage, UniformHandle sampler, TextureHandle h, uint32_t flags = UINT32_MAX);
void setUniform(UniformHandle h, const void* value, uint16_t num = 1);
void submit(ViewId id, ProgramHandle program, uint32_t depth = 0, uint8_t flags = 0xff);
}

#endif
