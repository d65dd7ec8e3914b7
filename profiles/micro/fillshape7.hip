// Micro-benchmark (tuning aid): the emit kernels' speed depends on which physical pages the output buffers got (DESIGN.md
// section 9). Is that a property of the BLOCKED wave -> address mapping (every wave streams its own long range: ~8192
// waves x 4 streams = tens of thousands of active pages) that a round-robin mapping (the resident waves cover one compact
// moving window) would not have? Several allocation sets in one process, both mappings on each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct V16 { uint32_t v[4]; };
struct V8 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) I9 { uint32_t a, b, c, d; uint16_t e; };

__device__ __forceinline__ void chunk(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t ch)
{
	const float2 p = in[ch * 64 + threadIdx.x];
	float x = p.x, y = p.y;
#pragma unroll
	for (int k = 0; k < 64; ++k) { x = x * 1.0001f + y; y = y * 0.9999f - x; }
	V16 q; q.v[0] = __float_as_uint(x); q.v[1] = __float_as_uint(y); q.v[2] = q.v[0] ^ 1; q.v[3] = q.v[1] ^ 1;
	*(V16*)(a + ch * 1024 + threadIdx.x * 16) = q;
	V8 r; r.v[0] = q.v[0]; r.v[1] = q.v[1];
	*(V8*)(b + ch * 512 + threadIdx.x * 8) = r;
	I9 s; s.a = q.v[0]; s.b = q.v[1]; s.c = q.v[2]; s.d = q.v[3]; s.e = (uint16_t)threadIdx.x;
	*(I9*)(c + ch * 1152 + threadIdx.x * 18) = s;
}

__global__ __launch_bounds__(64) void k_map(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t chunks, uint32_t run)
{
	if (run == 0) {
		const uint64_t per = chunks / gridDim.x;
		for (uint64_t i = 0; i < per; ++i) { chunk(a, b, c, in, (uint64_t)blockIdx.x * per + i); }
	} else {
		const uint64_t runs = chunks / run;
		for (uint64_t r = blockIdx.x; r < runs; r += gridDim.x) {
			for (uint32_t i = 0; i < run; ++i) { chunk(a, b, c, in, r * run + i); }
		}
	}
}

template<class F>
static float best_ms(F launch)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep) {
		(void)hipEventRecord(e0);
		launch();
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) { best = ms; }
	}
	return best;
}

int main()
{
	const int g = 32768;
	const uint64_t chunks = ((6ull << 30) / 2688) / (32768 * 16) * (32768 * 16);
	for (int set = 0; set < 6; ++set) {
		uint8_t *a, *b, *c; float2* in;
		// separate allocations like the real caller's pos / colour / index buffers + the library's polyline heap
		if (hipMalloc(&a, chunks * 1024 + 4096) != hipSuccess || hipMalloc(&b, chunks * 512 + 4096) != hipSuccess || hipMalloc(&c, chunks * 1152 + 4096) != hipSuccess || hipMalloc(&in, chunks * 512 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
		(void)hipMemset(in, 0, chunks * 512);
		float t0 = best_ms([&] { hipLaunchKernelGGL(k_map, dim3(g), dim3(64), 0, 0, a, b, c, in, chunks, 0u); });
		float t8 = best_ms([&] { hipLaunchKernelGGL(k_map, dim3(g), dim3(64), 0, 0, a, b, c, in, chunks, 8u); });
		float t1 = best_ms([&] { hipLaunchKernelGGL(k_map, dim3(g), dim3(64), 0, 0, a, b, c, in, chunks, 1u); });
		printf("allocation set %d: blocked %.3f ms (%.2f TB/s) | round-robin runs of 8 chunks %.3f ms (%.2f) | of 1 chunk %.3f ms (%.2f)\n", set, t0, (double)chunks * 2688 / t0 / 1e9,
		       t8, (double)chunks * 2688 / t8 / 1e9, t1, (double)chunks * 2688 / t1 / 1e9);
		fflush(stdout);
		// keep the set allocated: the next one gets other pages
	}
	return 0;
}
