// Micro-benchmark (tuning aid), follow-up of fillshape3: reads and stores interfere super-additively (1.07 GB read alone
// 0.19 ms, 6 GB of stores alone 1.0 ms, together 1.65 ms). Does batching a wave's reads (B chunks' vertices loaded
// back to back, then B chunks of stores) bring the sum back?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct V16 { uint32_t v[4]; };
struct V8 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) I9 { uint32_t a, b, c, d; uint16_t e; };

template<int MATH>
__device__ __forceinline__ void emit(uint8_t* a, uint8_t* b, uint8_t* c, uint64_t ch, float x, float y)
{
#pragma unroll
	for (int k = 0; k < MATH; ++k) { x = x * 1.0001f + y; y = y * 0.9999f - x; }
	const uint32_t q0 = __float_as_uint(x), q1 = __float_as_uint(y);
	V16 q; q.v[0] = q0; q.v[1] = q1; q.v[2] = q0 ^ 1; q.v[3] = q1 ^ 1;
	*(V16*)(a + ch * 1024 + threadIdx.x * 16) = q;
	V8 r; r.v[0] = q0; r.v[1] = q1;
	*(V8*)(b + ch * 512 + threadIdx.x * 8) = r;
	I9 s; s.a = q0; s.b = q1; s.c = q0 ^ 1; s.d = q1 ^ 1; s.e = (uint16_t)threadIdx.x;
	*(I9*)(c + ch * 1152 + threadIdx.x * 18) = s;
}

// B chunks per batch; PRE: the NEXT batch's loads are issued before this batch's stores (double buffering)
template<int B, bool PRE, int MATH>
__global__ __launch_bounds__(64) void k_burst(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t chunks)
{
	const uint64_t per = chunks / gridDim.x; // multiple of 16
	const uint64_t c0 = (uint64_t)blockIdx.x * per;
	float2 p[B], n[B];
	if (PRE) {
#pragma unroll
		for (int i = 0; i < B; ++i) { n[i] = in[(c0 + i) * 64 + threadIdx.x]; }
	}
	for (uint64_t k = 0; k < per; k += B) {
		if (PRE) {
#pragma unroll
			for (int i = 0; i < B; ++i) { p[i] = n[i]; }
			if (k + B < per) {
#pragma unroll
				for (int i = 0; i < B; ++i) { n[i] = in[(c0 + k + B + i) * 64 + threadIdx.x]; }
			}
		} else {
#pragma unroll
			for (int i = 0; i < B; ++i) { p[i] = in[(c0 + k + i) * 64 + threadIdx.x]; }
		}
#pragma unroll
		for (int i = 0; i < B; ++i) { emit<MATH>(a, b, c, c0 + k + i, p[i].x, p[i].y); }
	}
}

// the batch goes through LDS: one wave-wide copy of B*512 contiguous bytes with 16 B/lane loads, then per-chunk ds_read
template<int B, int MATH>
__global__ __launch_bounds__(64) void k_burst_lds(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t chunks)
{
	__shared__ float4 s[B * 32];
	const uint64_t per = chunks / gridDim.x;
	const uint64_t c0 = (uint64_t)blockIdx.x * per;
	for (uint64_t k = 0; k < per; k += B) {
		const float4* src = (const float4*)(in + (c0 + k) * 64);
		float4 t[B / 2];
#pragma unroll
		for (int i = 0; i < B / 2; ++i) { t[i] = src[i * 64 + threadIdx.x]; }
		__syncthreads();
#pragma unroll
		for (int i = 0; i < B / 2; ++i) { s[i * 64 + threadIdx.x] = t[i]; }
		__syncthreads();
#pragma unroll
		for (int i = 0; i < B; ++i) {
			const float2 p = ((const float2*)s)[i * 64 + threadIdx.x];
			emit<MATH>(a, b, c, c0 + k + i, p.x, p.y);
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

#define RUN(KERNEL, label) { float ms = best_ms([&] { hipLaunchKernelGGL((KERNEL), dim3(g), dim3(64), 0, 0, a, b, c, in, chunks); }); \
	printf("grid=%5d %-50s %.3f ms  write %.2f TB/s\n", g, label, ms, (double)chunks * 2688 / ms / 1e9); }

int main()
{
	const uint64_t bytes = 6ull << 30;
	uint8_t* buf; float2* in;
	const uint64_t chunks = (bytes / 2688) / (32768 * 16) * (32768 * 16);
	if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess || hipMalloc(&in, chunks * 512 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void)hipMemset(in, 0, chunks * 512);
	uint8_t* a = buf; uint8_t* b = buf + chunks * 1024 + 4096; uint8_t* c = buf + chunks * 1536 + 8192;
	for (int g : { 4096, 8192, 32768 }) {
		RUN((k_burst<1, false, 64>), "batch 1");
		RUN((k_burst<2, false, 64>), "batch 2");
		RUN((k_burst<4, false, 64>), "batch 4");
		RUN((k_burst<8, false, 64>), "batch 8");
		RUN((k_burst<16, false, 64>), "batch 16");
		RUN((k_burst<4, true, 64>), "batch 4, next batch loaded ahead");
		RUN((k_burst<8, true, 64>), "batch 8, next batch loaded ahead");
		RUN((k_burst<16, true, 64>), "batch 16, next batch loaded ahead");
		RUN((k_burst_lds<8, 64>), "batch 8 through LDS (16 B/lane loads)");
		RUN((k_burst_lds<16, 64>), "batch 16 through LDS (16 B/lane loads)");
	}
	return 0;
}
