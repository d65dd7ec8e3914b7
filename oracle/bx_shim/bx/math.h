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
// Only text / font-atlas code reaches these two (fontstash.h:1373 blur, stb_truetype): outside the parity scope.
inline float exp(float a) { return ::expf(a); }
inline float pow(float a, float b) { return ::powf(a, b); }
// View / projection set-up of vg::end (vg.cpp:1151-1153); the matrices only travel to bgfx::setViewTransform.
inline void mtxIdentity(float* m) { ::memset(m, 0, sizeof(float) * 16); m[0] = m[5] = m[10] = m[15] = 1.0f; }
inline void mtxOrtho(float* m, float l, float r, float b, float t, float n, float f, float offset, bool homogeneousNdc)
{
	const float aa = 2.0f / (r - l), bb = 2.0f / (t - b);
	const float cc = (homogeneousNdc ? 2.0f : 1.0f) / (f - n);
	const float dd = (l + r) / (l - r), ee = (t + b) / (b - t);
	const float ff = homogeneousNdc ? (n + f) / (n - f) : n / (n - f);
	::memset(m, 0, sizeof(float) * 16);
	m[0] = aa; m[5] = bb; m[10] = cc; m[12] = dd + offset; m[13] = ee; m[14] = ff; m[15] = 1.0f;
}
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
