// Minimal stand-in for bkaradzic/bx's <bx/bx.h>, just enough to compile the UNMODIFIED reference
// sources src/path.cpp, src/stroker.cpp, src/vg_util.cpp for the parity oracle (oracle/_ref).
// bx is an un-vendored, un-pinned dependency of the reference (README.md:91-93); this shim is test
// infrastructure, not product code.
#ifndef BX_SHIM_BX_H
#define BX_SHIM_BX_H

#include <stdint.h>
#include <stddef.h>
#include <string.h>

#if defined(__x86_64__) || defined(__i386__)
#define BX_CPU_X86 1
#else
#define BX_CPU_X86 0
#endif

#define BX_FORCE_INLINE inline __attribute__((always_inline))
#define BX_CONSTEXPR_FUNC constexpr
#define BX_ALIGN_DECL(_align, _decl) _decl __attribute__((aligned(_align)))
#define BX_ALIGN_DECL_16(_decl) BX_ALIGN_DECL(16, _decl)
#define BX_PRAGMA_DIAGNOSTIC_IGNORED_MSVC(_x)
#define BX_PRAGMA_DIAGNOSTIC_IGNORED_CLANG_GCC(_x)
#define BX_PRAGMA_DIAGNOSTIC_IGNORED_GCC(_x)
#define BX_PRAGMA_DIAGNOSTIC_IGNORED_CLANG(_x)
#define BX_PRAGMA_DIAGNOSTIC_PUSH()
#define BX_PRAGMA_DIAGNOSTIC_POP()
#define BX_COUNTOF(_x) (sizeof(_x) / sizeof((_x)[0]))
#define BX_FILE_LINE_LITERAL ""
#ifndef BX_CONFIG_SUPPORTS_THREADING
#define BX_CONFIG_SUPPORTS_THREADING 1
#endif
#ifndef BX_PLATFORM_EMSCRIPTEN
#define BX_PLATFORM_EMSCRIPTEN 0
#endif

namespace bx
{
template<typename... Args> inline void unusedArgs(Args&&...) {}
inline void memSet(void* dst, uint8_t ch, size_t n) { ::memset(dst, ch, n); }
inline void memCopy(void* dst, const void* src, size_t n) { ::memcpy(dst, src, n); }
inline void memMove(void* dst, const void* src, size_t n) { ::memmove(dst, src, n); }
inline int32_t memCmp(const void* a, const void* b, size_t n) { return ::memcmp(a, b, n); }
// strided rows -> packed (vg.cpp:2288, texture sub-rect upload)
inline void gather(void* dst, const void* src, uint32_t srcStride, uint32_t size, uint32_t num)
{
	uint8_t* d = (uint8_t*)dst; const uint8_t* s = (const uint8_t*)src;
	for (uint32_t i = 0; i < num; ++i) { ::memcpy(d, s, size); d += size; s += srcStride; }
}
template<typename T> inline constexpr bool isPowerOf2(T a) { return a && !(a & (a - 1)); }
}
#define BX_UNUSED(...) bx::unusedArgs(__VA_ARGS__)

#endif
