// Stand-in for <bx/debug.h> (only reached with VG_CONFIG_DEBUG). Test infrastructure only.
#ifndef BX_SHIM_DEBUG_H
#define BX_SHIM_DEBUG_H
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
namespace bx
{
inline void debugPrintf(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
inline void debugBreak() { abort(); }
}
#endif
