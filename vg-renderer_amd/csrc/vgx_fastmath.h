// vgx_fastmath.h -- correctly rounded 1/x and 1/sqrt(x) for the element kernels, in a handful of instructions.
//
// The arithmetic contract of the path (vgmath.h) is IEEE binary32 with correctly rounded + - * / sqrt. The compiler's
// generic sequences for `/` and `sqrtf` (-fhip-fp32-correctly-rounded-divide-sqrt) cost ~11 and ~15 VALU instructions
// because they must survive denormals, infinities and the whole exponent range (v_div_scale / v_div_fmas / v_div_fixup,
// range scaling around v_sqrt_f32). The element kernels only ever take
//     vec2Dir:              1 / sqrt(lenSqr)   with lenSqr in [1e-5, FLT_MAX]         (stroker.cpp:31-38)
//     calcExtrusionVector:  1 / cross          with |cross| in (0.01, ~1]            (stroker.cpp:40-53)
// i.e. normal numbers far from both ends of the exponent range, where one hardware estimate (v_rcp_f32 / v_sqrt_f32, 1 ulp)
// plus FMA residual steps gives the correctly rounded result. "Gives" is not argued, it is CHECKED: the functions below are
// compared with the compiler's correctly rounded `/` and sqrtf over EVERY float of their domain by
// tests/native/exact_math_test.hip (run by tests/test_gpu_exact_math.py on the GPU). The FMAs are explicit builtins: they
// are part of these algorithms, not contractions of the reference's expressions (-ffp-contract=off stays in force).
#ifndef VGX_FASTMATH_H
#define VGX_FASTMATH_H

#include <hip/hip_runtime.h>

// Refinement steps: the smallest counts for which tests/native/exact_math_test.hip finds no mismatch over the whole domain
// (measured on gfx950: one step leaves RN(sqrt) wrong for the all-ones mantissa and RN(1 / RN(sqrt)) for 5 mantissas).
#ifndef VGX_FASTMATH_RCP_STEPS
#define VGX_FASTMATH_RCP_STEPS 1
#endif
#ifndef VGX_FASTMATH_SQRT_STEPS
#define VGX_FASTMATH_SQRT_STEPS 2
#endif
#ifndef VGX_FASTMATH_RSQ_STEPS
#define VGX_FASTMATH_RSQ_STEPS 2
#endif

// RN(1 / b) for normal b with 2^-100 <= |b| <= 2^100.
__device__ __forceinline__ float vgx_rcp_rn(float b)
{
	float y = __builtin_amdgcn_rcpf(b);
	float e = __builtin_fmaf(-b, y, 1.0f);
	y = __builtin_fmaf(e, y, y);
#if VGX_FASTMATH_RCP_STEPS >= 2
	e = __builtin_fmaf(-b, y, 1.0f);
	y = __builtin_fmaf(e, y, y);
#endif
	return y;
}

// RN(sqrt(x)) for normal x in [2^-100, 2^100].
__device__ __forceinline__ float vgx_sqrt_rn(float x)
{
	const float s0 = __builtin_amdgcn_sqrtf(x);
	const float h = 0.5f * __builtin_amdgcn_rcpf(s0);
	const float r = __builtin_fmaf(-s0, s0, x);
	float s = __builtin_fmaf(r, h, s0);
#if VGX_FASTMATH_SQRT_STEPS >= 2
	s = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
#endif
	return s;
}

// RN(1 / RN(sqrt(x))): what vgm_rsqrt computes (1.0f / sqrtf(x)), for normal x in [2^-100, 2^100].
__device__ __forceinline__ float vgx_rsqrt_rn(float x)
{
	const float s0 = __builtin_amdgcn_sqrtf(x);
	const float y0 = __builtin_amdgcn_rcpf(s0);
	const float r = __builtin_fmaf(-s0, s0, x);
	float s = __builtin_fmaf(r, 0.5f * y0, s0);   // RN(sqrt(x))
#if VGX_FASTMATH_SQRT_STEPS >= 2
	s = __builtin_fmaf(__builtin_fmaf(-s, s, x), 0.5f * y0, s);
#endif
	float e = __builtin_fmaf(-s, y0, 1.0f);
	float y = __builtin_fmaf(e, y0, y0);
#if VGX_FASTMATH_RSQ_STEPS >= 2
	e = __builtin_fmaf(-s, y, 1.0f);
	y = __builtin_fmaf(e, y, y);
#endif
	return y;
}

#endif
