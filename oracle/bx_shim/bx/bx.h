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

namespace bx
{
template<typename... Args> inline void unusedArgs(Args&&...) {}
inline void memSet(void* dst, uint8_t ch, size_t n) { ::memset(dst, ch, n); }
inline void memCopy(void* dst, const void* src, size_t n) { ::memcpy(dst, src, n); }
inline void memMove(void* dst, const void* src, size_t n) { ::memmove(dst, src, n); }
}
#define BX_UNUSED(...) bx::unusedArgs(__VA_ARGS__)

#endif
