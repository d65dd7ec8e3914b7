// vgmath.h -- pinned float32 arithmetic shared by the HIP kernels, the host side and the oracle.
//
// WHY THIS FILE EXISTS
// The reference's hot path (src/path.cpp, src/stroker.cpp) takes its transcendentals from `bx`
// (bx::acos path.cpp:307,602,654 stroker.cpp:1013,1398; bx::atan2 stroker.cpp:1140,1588,...;
// bx::cos/sin stroker.cpp:1161,1626,...; bx::tan path.cpp:246; bx::rsqrt stroker.cpp:36).
// bx is not vendored in the reference and no version is pinned, so that arithmetic is *unpinned*.
// Vertex COUNTS (and therefore every uint16 index) depend on it through
//   numPointsHalfCircle = max(2, ceil(pi / (2*acos(...))))  and  numArcPoints = max(2, (u32)(dA/da)),
// so GPU and CPU must evaluate these functions bit-identically. This header *defines* them using only
// + - * / sqrt floor, which are correctly rounded on x86-64 and on gfx950 when both sides are compiled
// with -ffp-contract=off (no FMA contraction) and without fast-math. Nothing here may call libm/ocml
// transcendentals.
//
// Polynomial coefficients are the classic single-precision Cephes minimax sets (public constants).
#ifndef VGMATH_H
#define VGMATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define VGM_FN __host__ __device__ __forceinline__
#else
#define VGM_FN static inline
#endif

#define VGM_PI      3.14159265358979323846f
#define VGM_PI2     6.28318530717958647692f
#define VGM_PIHALF  1.57079632679489661923f
#define VGM_PIQUART 0.78539816339744830962f
#define VGM_EPSILON 1e-5f /* VG_EPSILON, include/vg/vg.h:88 */

VGM_FN float vgm_abs(float a) { return fabsf(a); }
VGM_FN float vgm_sign(float a) { return (float)((0.0f < a) - (0.0f > a)); } /* 0 -> 0 */
VGM_FN float vgm_min(float a, float b) { return a < b ? a : b; }
VGM_FN float vgm_max(float a, float b) { return a > b ? a : b; }
VGM_FN uint32_t vgm_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
VGM_FN float vgm_floor(float a) { return floorf(a); }
VGM_FN float vgm_ceil(float a) { return ceilf(a); }
VGM_FN float vgm_sqrt(float a) { return sqrtf(a); }

/* 1/sqrt(x): one correctly rounded sqrt followed by one correctly rounded divide. */
VGM_FN float vgm_rsqrt(float a) { return 1.0f / sqrtf(a); }

/* Cody-Waite reduction of x to r in [-pi/4, pi/4] and a quadrant q = 0..3, x = r + q*pi/2 (mod 2pi). */
VGM_FN float vgm_reduce_pio2(float x, int* q)
{
	const float fx = floorf(x * 0.636619772367581343f + 0.5f);
	float r = x - fx * 1.5703125f;
	r = r - fx * 4.837512969970703125e-4f;
	r = r - fx * 7.54978995489188e-8f;
	*q = ((int)fx) & 3;
	return r;
}

VGM_FN float vgm_sin_poly(float r)
{
	const float z = r * r;
	return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
}

VGM_FN float vgm_cos_poly(float r)
{
	const float z = r * r;
	return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
}

VGM_FN void vgm_sincos(float x, float* s, float* c)
{
	int q;
	const float r = vgm_reduce_pio2(x, &q);
	const float sr = vgm_sin_poly(r);
	const float cr = vgm_cos_poly(r);
	const float ss = (q & 1) ? cr : sr;
	const float cc = (q & 1) ? sr : cr;
	*s = (q & 2) ? -ss : ss;
	*c = ((q + 1) & 2) ? -cc : cc;
}

VGM_FN float vgm_sin(float x) { float s, c; vgm_sincos(x, &s, &c); return s; }
VGM_FN float vgm_cos(float x) { float s, c; vgm_sincos(x, &s, &c); return c; }
VGM_FN float vgm_tan(float x) { float s, c; vgm_sincos(x, &s, &c); return s / c; }

VGM_FN float vgm_asin_poly(float x)
{
	const float z = x * x;
	return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
}

/* acos on [-1,1]; arguments outside are clamped (the reference feeds (s*r)/(s*r+tol) in [0,1) and
 * dot products of unit vectors). */
VGM_FN float vgm_acos(float x)
{
	if (x > 1.0f) { x = 1.0f; }
	if (x < -1.0f) { x = -1.0f; }
	if (x > 0.5f) {
		const float z = 0.5f * (1.0f - x);
		return 2.0f * vgm_asin_poly(sqrtf(z));
	}
	if (x < -0.5f) {
		const float z = 0.5f * (1.0f + x);
		return VGM_PI - 2.0f * vgm_asin_poly(sqrtf(z));
	}
	return VGM_PIHALF - vgm_asin_poly(x);
}

VGM_FN float vgm_atan(float xin)
{
	const float x = fabsf(xin);
	float y, t;
	if (x > 2.414213562373095f) {
		y = VGM_PIHALF;
		t = -(1.0f / x);
	} else if (x > 0.4142135623730950f) {
		y = VGM_PIQUART;
		t = (x - 1.0f) / (x + 1.0f);
	} else {
		y = 0.0f;
		t = x;
	}
	const float z = t * t;
	y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * t + t);
	return xin < 0.0f ? -y : y;
}

/* atan2 with fully specified corner cases (+-0 compare equal to 0). */
VGM_FN float vgm_atan2(float y, float x)
{
	if (x == 0.0f) {
		if (y > 0.0f) { return VGM_PIHALF; }
		if (y < 0.0f) { return -VGM_PIHALF; }
		return 0.0f;
	}
	if (y == 0.0f) {
		return x > 0.0f ? 0.0f : VGM_PI;
	}
	const float a = vgm_atan(y / x);
	if (x > 0.0f) { return a; }
	return y > 0.0f ? a + VGM_PI : a - VGM_PI;
}

#endif // VGMATH_H
