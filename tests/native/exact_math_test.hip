// Exhaustive check of csrc/vgx_fastmath.h against the compiler's correctly rounded `/` and sqrtf: EVERY binary32 value of
// the functions' domains (both signs for the reciprocal). Prints one line per function: "<name> mismatches=<n> of <total>".
// Built and run by tests/test_gpu_exact_math.py; exit code 1 when any mismatch is found.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include "../../vg-renderer_amd/csrc/vgx_fastmath.h"

// MODE 0: rcp, 1: sqrt, 2: rsqrt. bits in [lo, hi).
template<int MODE>
__global__ void k_check(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* firstBad)
{
	unsigned long long local = 0;
	for (uint64_t b = (uint64_t)lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * blockDim.x) {
		const float x = __uint_as_float((uint32_t)b);
		float want, got;
		if (MODE == 0) {
			want = 1.0f / x; got = vgx_rcp_rn(x);
			const float xn = -x;
			const float wn = 1.0f / xn, gn = vgx_rcp_rn(xn);
			if (__float_as_uint(wn) != __float_as_uint(gn)) { ++local; atomicMin(firstBad, (uint32_t)b); }
		} else if (MODE == 1) {
			want = sqrtf(x); got = vgx_sqrt_rn(x);
		} else {
			want = 1.0f / sqrtf(x); got = vgx_rsqrt_rn(x);
		}
		if (__float_as_uint(want) != __float_as_uint(got)) { ++local; atomicMin(firstBad, (uint32_t)b); }
	}
	if (local) { atomicAdd(bad, local); }
}

template<int MODE>
static int run(const char* name, float lo, float hi)
{
	unsigned long long* bad; uint32_t* first;
	(void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 4);
	(void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0xFF, 4);
	uint32_t l, h;
	memcpy(&l, &lo, 4); memcpy(&h, &hi, 4);
	hipLaunchKernelGGL(k_check<MODE>, dim3(8192), dim3(256), 0, 0, l, h, bad, first);
	unsigned long long nb = 0; uint32_t fb = 0;
	(void)hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
	(void)hipMemcpy(&fb, first, 4, hipMemcpyDeviceToHost);
	printf("%s mismatches=%llu of %llu", name, nb, (unsigned long long)(h - l) * (MODE == 0 ? 2 : 1));
	if (nb) { float f; memcpy(&f, &fb, 4); printf(" first=0x%08x (%g)", fb, f); }
	printf("\n");
	(void)hipFree(bad); (void)hipFree(first);
	return nb ? 1 : 0;
}

int main()
{
	int rc = 0;
	rc |= run<0>("rcp_rn", ldexpf(1.0f, -100), ldexpf(1.0f, 100));
	rc |= run<1>("sqrt_rn", ldexpf(1.0f, -100), ldexpf(1.0f, 100));
	rc |= run<2>("rsqrt_rn", ldexpf(1.0f, -100), ldexpf(1.0f, 100));
	return rc;
}
