// Micro-benchmark (tuning aid): k_fill's access mix with the reads and the stores PHASED device-wide: every wave loads
// B chunks of vertices into LDS, all waves meet at a grid barrier, every wave stores its B chunks' output, barrier.
// Question: is the read-beside-write interference (1.07 GB read alone 0.19 ms, 6 GB of stores alone 1.0 ms, mixed
// 1.4-1.65 ms) avoidable by separating the two in time?
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

__device__ __forceinline__ void grid_barrier(unsigned int* ctr, unsigned int target)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		// bounded: a barrier that cannot complete (not all workgroups resident) gives up instead of hanging the device
		for (uint32_t spin = 0; spin < (1u << 20) && __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin) { __builtin_amdgcn_s_sleep(4); }
		if (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) { __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	}
	__syncthreads();
}

// B chunks per phase and wave; PHASED = with the two grid barriers
template<int B, bool PHASED, int MATH>
__global__ __launch_bounds__(64) void k_phased(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t chunks, unsigned int* ctr)
{
	__shared__ float2 s[B * 64];
	const uint64_t per = chunks / gridDim.x;
	const uint64_t c0 = (uint64_t)blockIdx.x * per;
	unsigned int phase = 0;
	for (uint64_t k = 0; k < per; k += B) {
		const float4* src = (const float4*)(in + (c0 + k) * 64);
#pragma unroll
		for (int i = 0; i < B / 2; ++i) { ((float4*)s)[i * 64 + threadIdx.x] = src[i * 64 + threadIdx.x]; }
		__syncthreads();
		if (PHASED) { grid_barrier(ctr, ++phase * gridDim.x); }
#pragma unroll 4
		for (int i = 0; i < B; ++i) {
			const float2 p = s[i * 64 + threadIdx.x];
			emit<MATH>(a, b, c, c0 + k + i, p.x, p.y);
		}
		if (PHASED) { __builtin_amdgcn_s_waitcnt(0x0F70); grid_barrier(ctr, ++phase * gridDim.x); }
		__syncthreads();
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

#define RUN(B, PH, g, label) { float ms = best_ms([&] { (void)hipMemsetAsync(ctr, 0, 8, 0); hipLaunchKernelGGL((k_phased<B, PH, 64>), dim3(g), dim3(64), 0, 0, a, b, c, in, chunks, ctr); }); \
	unsigned int h[2]; (void)hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost); \
	printf("grid=%5d B=%3d %-28s %.3f ms  write %.2f TB/s  (barrier give-ups %u)\n", g, B, label, ms, (double)chunks * 2688 / ms / 1e9, h[1]); fflush(stdout); }

int main()
{
	const uint64_t bytes = 6ull << 30;
	uint8_t* buf; float2* in; unsigned int* ctr;
	const uint64_t chunks = (bytes / 2688) / (4096 * 64) * (4096 * 64);
	if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess || hipMalloc(&in, chunks * 512 + 4096) != hipSuccess || hipMalloc(&ctr, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void)hipMemset(in, 0, chunks * 512);
	uint8_t* a = buf; uint8_t* b = buf + chunks * 1024 + 4096; uint8_t* c = buf + chunks * 1536 + 8192;
	RUN(16, false, 4096, "unphased, LDS batch");
	RUN(64, false, 1024, "unphased, LDS batch");
	RUN(64, false, 2048, "unphased, LDS batch");
	RUN(16, true, 1024, "phased (grid barriers)");
	RUN(32, true, 1024, "phased (grid barriers)");
	RUN(64, true, 1024, "phased (grid barriers)");
	RUN(16, true, 2048, "phased (grid barriers)");
	RUN(32, true, 2048, "phased (grid barriers)");
	RUN(64, true, 2048, "phased (grid barriers)");
	RUN(16, true, 4096, "phased (grid barriers)");
	RUN(32, true, 4096, "phased (grid barriers)");
	return 0;
}
