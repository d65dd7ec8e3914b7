// Micro-benchmark (tuning aid): what limits a k_fill-shaped kernel? Three contiguous output streams (16 + 8 + 18 B per
// lane per iteration, like pos / colour / index of convexFillAA), optionally an 8 B/lane input stream and a tunable
// amount of dependent float math per iteration (k_fill executes ~270 VALU instructions per 64-element chunk).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct V16 { uint32_t v[4]; };
struct V8 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) I9 { uint32_t a, b, c, d; uint16_t e; };

// same as k_shape<MATH, true> but the input stream wraps inside `wrap` float2 elements (a working set that stays in the
// 256 MB Infinity Cache): does the read stream still cost write bandwidth when it does not come from HBM?
template<int MATH>
__global__ __launch_bounds__(64) void k_shape_cached(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t iters, uint64_t wrap)
{
	const uint64_t perWave = iters / gridDim.x;
	uint8_t* pa = a + (uint64_t)blockIdx.x * perWave * 1024;
	uint8_t* pb = b + (uint64_t)blockIdx.x * perWave * 512;
	uint8_t* pc = c + (uint64_t)blockIdx.x * perWave * 1152;
	const uint64_t base = (uint64_t)blockIdx.x * perWave * 64;
	for (uint64_t it = 0; it < perWave; ++it) {
		const float2 p = in[(base + it * 64 + threadIdx.x) % wrap];
		float x = p.x, y = p.y;
#pragma unroll
		for (int k = 0; k < MATH; ++k) {
			x = x * 1.0001f + y;
			y = y * 0.9999f - x;
		}
		V16 q; q.v[0] = __float_as_uint(x); q.v[1] = __float_as_uint(y); q.v[2] = q.v[0] ^ 1; q.v[3] = q.v[1] ^ 1;
		*(V16*)(pa + it * 1024 + threadIdx.x * 16) = q;
		V8 r; r.v[0] = q.v[0]; r.v[1] = q.v[1];
		*(V8*)(pb + it * 512 + threadIdx.x * 8) = r;
		I9 s; s.a = q.v[0]; s.b = q.v[1]; s.c = q.v[2]; s.d = q.v[3]; s.e = (uint16_t)threadIdx.x;
		*(I9*)(pc + it * 1152 + threadIdx.x * 18) = s;
	}
}

template<int MATH, bool READ>
__global__ __launch_bounds__(64) void k_shape(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t iters)
{
	const uint64_t perWave = iters / gridDim.x;
	uint8_t* pa = a + (uint64_t)blockIdx.x * perWave * 1024;
	uint8_t* pb = b + (uint64_t)blockIdx.x * perWave * 512;
	uint8_t* pc = c + (uint64_t)blockIdx.x * perWave * 1152;
	const float2* pin = in + (uint64_t)blockIdx.x * perWave * 64;
	for (uint64_t it = 0; it < perWave; ++it) {
		float2 p = make_float2((float)threadIdx.x, (float)it);
		if (READ) { p = pin[it * 64 + threadIdx.x]; }
		float x = p.x, y = p.y;
#pragma unroll
		for (int k = 0; k < MATH; ++k) { // 4 dependent-ish VALU ops per step
			x = x * 1.0001f + y;
			y = y * 0.9999f - x;
		}
		V16 q; q.v[0] = __float_as_uint(x); q.v[1] = __float_as_uint(y); q.v[2] = q.v[0] ^ 1; q.v[3] = q.v[1] ^ 1;
		*(V16*)(pa + it * 1024 + threadIdx.x * 16) = q;
		V8 r; r.v[0] = q.v[0]; r.v[1] = q.v[1];
		*(V8*)(pb + it * 512 + threadIdx.x * 8) = r;
		I9 s; s.a = q.v[0]; s.b = q.v[1]; s.c = q.v[2]; s.d = q.v[3]; s.e = (uint16_t)threadIdx.x;
		*(I9*)(pc + it * 1152 + threadIdx.x * 18) = s;
	}
}

// the same work with the NEXT iteration's load issued before this iteration's stores (vmcnt counts loads and stores
// on gfx9: a load issued after the stores cannot be waited for without waiting for the stores' write acknowledgements)
template<int MATH>
__global__ __launch_bounds__(64) void k_shape_prefetch(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t iters)
{
	const uint64_t perWave = iters / gridDim.x;
	uint8_t* pa = a + (uint64_t)blockIdx.x * perWave * 1024;
	uint8_t* pb = b + (uint64_t)blockIdx.x * perWave * 512;
	uint8_t* pc = c + (uint64_t)blockIdx.x * perWave * 1152;
	const float2* pin = in + (uint64_t)blockIdx.x * perWave * 64;
	float2 p = pin[threadIdx.x];
	for (uint64_t it = 0; it < perWave; ++it) {
		float2 pn = p;
		if (it + 1 < perWave) { pn = pin[(it + 1) * 64 + threadIdx.x]; }
		float x = p.x, y = p.y;
#pragma unroll
		for (int k = 0; k < MATH; ++k) {
			x = x * 1.0001f + y;
			y = y * 0.9999f - x;
		}
		V16 q; q.v[0] = __float_as_uint(x); q.v[1] = __float_as_uint(y); q.v[2] = q.v[0] ^ 1; q.v[3] = q.v[1] ^ 1;
		*(V16*)(pa + it * 1024 + threadIdx.x * 16) = q;
		V8 r; r.v[0] = q.v[0]; r.v[1] = q.v[1];
		*(V8*)(pb + it * 512 + threadIdx.x * 8) = r;
		I9 s; s.a = q.v[0]; s.b = q.v[1]; s.c = q.v[2]; s.d = q.v[3]; s.e = (uint16_t)threadIdx.x;
		*(I9*)(pc + it * 1152 + threadIdx.x * 18) = s;
		p = pn;
	}
}

template<int MATH, bool READ>
static void run(const char* name, int grid, uint8_t* buf, const float2* in, uint64_t iters)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep) {
		hipEventRecord(e0);
		hipLaunchKernelGGL((k_shape<MATH, READ>), dim3(grid), dim3(64), 0, 0, buf, buf + iters * 1024 + 4096, buf + iters * 1536 + 8192, in, iters);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) { best = ms; }
	}
	printf("%-40s grid=%d: %.3f ms  write %.2f TB/s\n", name, grid, best, (double)iters * 2688 / best / 1e9);
}

int main()
{
	const uint64_t bytes = 6ull << 30;
	uint8_t* buf; float2* in;
	const int g = 32768;
	const uint64_t iters = (bytes / 2688) / g * g;
	if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess || hipMalloc(&in, iters * 512) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMemset(in, 0, iters * 512);
	run<0, false>("stores only", g, buf, in, iters);
	run<0, true>("+ 8 B/lane read", g, buf, in, iters);
	run<16, true>("+ read + 64 VALU", g, buf, in, iters);
	run<32, true>("+ read + 128 VALU", g, buf, in, iters);
	run<64, true>("+ read + 256 VALU", g, buf, in, iters);
	run<96, true>("+ read + 384 VALU", g, buf, in, iters);
	run<64, false>("256 VALU, no read", g, buf, in, iters);
	for (uint64_t mb : { 16ull, 64ull, 128ull, 512ull }) {
		hipEvent_t e0, e1;
		hipEventCreate(&e0); hipEventCreate(&e1);
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			hipEventRecord(e0);
			hipLaunchKernelGGL((k_shape_cached<64>), dim3(g), dim3(64), 0, 0, buf, buf + iters * 1024 + 4096, buf + iters * 1536 + 8192, in, iters, (mb << 20) / 8);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			if (ms < best) { best = ms; }
		}
		printf("read wraps in %4llu MB + 256 VALU          grid=%d: %.3f ms  write %.2f TB/s\n", (unsigned long long)mb, g, best, (double)iters * 2688 / best / 1e9);
	}
	{
		hipEvent_t e0, e1;
		hipEventCreate(&e0); hipEventCreate(&e1);
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			hipEventRecord(e0);
			hipLaunchKernelGGL((k_shape_prefetch<64>), dim3(g), dim3(64), 0, 0, buf, buf + iters * 1024 + 4096, buf + iters * 1536 + 8192, in, iters);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			if (ms < best) { best = ms; }
		}
		printf("%-40s grid=%d: %.3f ms  write %.2f TB/s\n", "read one iteration ahead + 256 VALU", g, best, (double)iters * 2688 / best / 1e9);
	}
	return 0;
}
