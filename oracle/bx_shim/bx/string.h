// Stand-in for <bx/string.h> (strLen only; vg.cpp:1910, 2132, 2920, 2941, 4188). Test infrastructure only.
#ifndef BX_SHIM_STRING_H
#define BX_SHIM_STRING_H
#include "bx.h"
namespace bx
{
inline int32_t strLen(const char* s, int32_t max = INT32_MAX) { if (!s) { return 0; } int32_t n = 0; while (n < max && s[n]) { ++n; } return n; }
}
#endif
