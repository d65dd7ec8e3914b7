// Micro-benchmark (tuning aid): does the wave -> address mapping limit a k_fill-shaped kernel (8 B/lane read, 16 + 8 +
// 18 B/lane written per 64-element chunk)? BLOCKED = every wave streams its own long contiguous range (what k_fill /
// k_stroke did in round 1: ~8192 resident waves = ~32 k independent read / write fronts advancing 0.5-1 KB at a time);
// RUN=R = waves take runs of R consecutive chunks round-robin (run r belongs to wave r % grid), so the resident waves
// cover one compact, moving window of the streams.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct V16 { uint32_t v[4]; };
struct V8 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) I9 { uint32_t a, b, c, d; uint16_t e; };

template<bool READ, int MATH>
__device__ __forceinline__ void chunk(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t ch)
{
	float2 p = make_float2((float)threadIdx.x, (float)ch);
	if (READ) { p = in[ch * 64 + threadIdx.x]; }
	float x = p.x, y = p.y;
#pragma unroll
	for (int k = 0; k < MATH; ++k) { x = x * 1.0001f + y; y = y * 0.9999f - x; }
	V16 q; q.v[0] = __float_as_uint(x); q.v[1] = __float_as_uint(y); q.v[2] = q.v[0] ^ 1; q.v[3] = q.v[1] ^ 1;
	*(V16*)(a + ch * 1024 + threadIdx.x * 16) = q;
	V8 r; r.v[0] = q.v[0]; r.v[1] = q.v[1];
	*(V8*)(b + ch * 512 + threadIdx.x * 8) = r;
	I9 s; s.a = q.v[0]; s.b = q.v[1]; s.c = q.v[2]; s.d = q.v[3]; s.e = (uint16_t)threadIdx.x;
	*(I9*)(c + ch * 1152 + threadIdx.x * 18) = s;
}

// RUN == 0: blocked
template<bool READ, int MATH>
__global__ __launch_bounds__(64) void k_map(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t chunks, uint32_t run)
{
	if (run == 0) {
		const uint64_t per = chunks / gridDim.x;
		for (uint64_t i = 0; i < per; ++i) { chunk<READ, MATH>(a, b, c, in, (uint64_t)blockIdx.x * per + i); }
	} else {
		const uint64_t runs = chunks / run;
		for (uint64_t r = blockIdx.x; r < runs; r += gridDim.x) {
			for (uint32_t i = 0; i < run; ++i) { chunk<READ, MATH>(a, b, c, in, r * run + i); }
		}
	}
}

// burst: the run's reads are all issued before its stores (RUN fixed at 4)
template<int MATH>
__global__ __launch_bounds__(64) void k_burst4(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t chunks, uint32_t blocked)
{
	const uint64_t runs = chunks / 4;
	const uint64_t per = runs / gridDim.x;
	for (uint64_t k = 0; k < per; ++k) {
		const uint64_t r = blocked ? (uint64_t)blockIdx.x * per + k : k * gridDim.x + blockIdx.x;
		float2 p[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) { p[i] = in[(r * 4 + i) * 64 + threadIdx.x]; }
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const uint64_t ch = r * 4 + i;
			float x = p[i].x, y = p[i].y;
#pragma unroll
			for (int m = 0; m < MATH; ++m) { x = x * 1.0001f + y; y = y * 0.9999f - x; }
			V16 q; q.v[0] = __float_as_uint(x); q.v[1] = __float_as_uint(y); q.v[2] = q.v[0] ^ 1; q.v[3] = q.v[1] ^ 1;
			*(V16*)(a + ch * 1024 + threadIdx.x * 16) = q;
			V8 rr; rr.v[0] = q.v[0]; rr.v[1] = q.v[1];
			*(V8*)(b + ch * 512 + threadIdx.x * 8) = rr;
			I9 s; s.a = q.v[0]; s.b = q.v[1]; s.c = q.v[2]; s.d = q.v[3]; s.e = (uint16_t)threadIdx.x;
			*(I9*)(c + ch * 1152 + threadIdx.x * 18) = s;
		}
	}
}

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ s, float4* __restrict__ d, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { d[i] = s[i]; }
}

template<class F>
static float best_ms(F launch)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep) {
		hipEventRecord(e0);
		launch();
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) { best = ms; }
	}
	return best;
}

int main()
{
	const uint64_t bytes = 6ull << 30;
	uint8_t* buf; float2* in;
	const uint64_t chunks = (bytes / 2688) / (32768 * 16) * (32768 * 16);
	if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess || hipMalloc(&in, chunks * 512) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMemset(in, 0, chunks * 512);
	uint8_t* a = buf; uint8_t* b = buf + chunks * 1024 + 4096; uint8_t* c = buf + chunks * 1536 + 8192;
	{
		const uint64_t n = (2ull << 30) / 16;
		float ms = best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const float4*)buf, (float4*)(buf + (3ull << 30)), n); });
		printf("float4 copy 2 GB -> 2 GB: %.3f ms  read+write %.2f TB/s\n", ms, 2.0 * n * 16 / ms / 1e9);
	}
	for (int g : { 8192, 32768 }) {
		for (uint32_t run : { 0u, 1u, 2u, 4u, 8u, 16u }) {
			float w = best_ms([&] { hipLaunchKernelGGL((k_map<false, 64>), dim3(g), dim3(64), 0, 0, a, b, c, in, chunks, run); });
			float rw = best_ms([&] { hipLaunchKernelGGL((k_map<true, 64>), dim3(g), dim3(64), 0, 0, a, b, c, in, chunks, run); });
			printf("grid=%5d run=%2u (0=blocked): stores only %.3f ms %.2f TB/s | + 8 B/lane read %.3f ms  write %.2f TB/s\n", g, run, w, (double)chunks * 2688 / w / 1e9, rw, (double)chunks * 2688 / rw / 1e9);
		}
		for (uint32_t blocked : { 1u, 0u }) {
			float rw = best_ms([&] { hipLaunchKernelGGL((k_burst4<64>), dim3(g), dim3(64), 0, 0, a, b, c, in, chunks, blocked); });
			printf("grid=%5d burst4 %s: %.3f ms  write %.2f TB/s\n", g, blocked ? "blocked" : "interleaved", rw, (double)chunks * 2688 / rw / 1e9);
		}
	}
	return 0;
}
