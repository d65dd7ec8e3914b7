// Stand-in for <bx/math.h>. Every transcendental forwards to vgmath.h, the repo's pinned float32
// arithmetic (see that header for why): this is what makes the compiled reference a bit-exact
// oracle for the HIP kernels. With -DVGO_SHIM_LIBM the transcendentals forward to glibc instead;
// that build only exists to MEASURE how sensitive vertex counts are to the unpinned bx arithmetic.
#ifndef BX_SHIM_MATH_H
#define BX_SHIM_MATH_H

#include "bx.h"
#include "vgmath.h"
#include <math.h>

namespace bx
{
constexpr float kPi = VGM_PI;
constexpr float kPi2 = VGM_PI2;
constexpr float kPiHalf = VGM_PIHALF;
constexpr float kPiQuarter = VGM_PIQUART;

inline float abs(float a) { return vgm_abs(a); }
inline float sign(float a) { return vgm_sign(a); }
inline float floor(float a) { return vgm_floor(a); }
inline float ceil(float a) { return vgm_ceil(a); }
inline float sqrt(float a) { return vgm_sqrt(a); }
inline float square(float a) { return a * a; }
inline float mod(float a, float b) { return ::fmodf(a, b); }
#ifdef VGO_SHIM_LIBM
inline float rsqrt(float a) { return 1.0f / ::sqrtf(a); }
inline float cos(float a) { return ::cosf(a); }
inline float sin(float a) { return ::sinf(a); }
inline float tan(float a) { return ::tanf(a); }
inline float acos(float a) { return ::acosf(a); }
inline float atan2(float y, float x) { return ::atan2f(y, x); }
#else
inline float rsqrt(float a) { return vgm_rsqrt(a); }
inline float cos(float a) { return vgm_cos(a); }
inline float sin(float a) { return vgm_sin(a); }
inline float tan(float a) { return vgm_tan(a); }
inline float acos(float a) { return vgm_acos(a); }
inline float atan2(float y, float x) { return vgm_atan2(y, x); }
#endif

template<typename T> inline T min(const T& a, const T& b) { return a < b ? a : b; }
template<typename T> inline T max(const T& a, const T& b) { return a > b ? a : b; }
template<typename T> inline T min(const T& a, const T& b, const T& c) { return min(min(a, b), c); }
template<typename T> inline T max(const T& a, const T& b, const T& c) { return max(max(a, b), c); }
template<typename T> inline T clamp(const T& a, const T& lo, const T& hi) { return max(min(a, hi), lo); }
inline uint32_t uint32_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
inline uint32_t uint32_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
}

#endif
