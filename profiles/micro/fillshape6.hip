// Micro-benchmark (tuning aid): k_fill's access mix with the vertex stream prefetched by LDS-DMA (global_load_lds_dword,
// issued from inline asm so that hipcc neither counts it nor waits for it) into a ring of chunk slots, arrival detected
// by CONTENT: the slot is filled with a NaN sentinel before the request and polled with ds_read when the chunk is
// emitted. The point: no s_waitcnt vmcnt ever stands between a wave's stores and its next loads (gfx9 has one counter
// for both; the compiler must assume vmcnt(0) as soon as a store sits under a divergent branch).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct V16 { uint32_t v[4]; };
struct V8 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) I9 { uint32_t a, b, c, d; uint16_t e; };
#define SENT 0x7FC0DEADu

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

// one dword per lane from gsrc to LDS byte address lds_dst + lane * 4 (lds_dst wave-uniform)
__device__ __forceinline__ void glds_dword(const void* gsrc, uint32_t lds_dst)
{
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template<int RING, int AHEAD, int MATH>
__global__ __launch_bounds__(64) void k_ring(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t chunks, unsigned int* giveups)
{
	__shared__ uint32_t sx[RING * 64];
	__shared__ uint32_t sy[RING * 64];
	const uint64_t per = chunks / gridDim.x;
	const uint64_t c0 = (uint64_t)blockIdx.x * per;
	const int lane = threadIdx.x;
	const uint32_t bx = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)sx;
	const uint32_t by = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)sy;
	auto request = [&](uint64_t k) {
		const uint32_t slot = (uint32_t)(k % RING);
		sx[slot * 64 + lane] = SENT; sy[slot * 64 + lane] = SENT;
		__builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the sentinels are in LDS before the DMA can land
		const float2* src = in + (c0 + k) * 64 + lane;
		glds_dword(&src->x, __builtin_amdgcn_readfirstlane(bx + slot * 256));
		glds_dword(&src->y, __builtin_amdgcn_readfirstlane(by + slot * 256));
	};
	for (uint64_t k = 0; k < (uint64_t)AHEAD && k < per; ++k) { request(k); }
	for (uint64_t k = 0; k < per; ++k) {
		if (k + AHEAD < per) { request(k + AHEAD); }
		const uint32_t slot = (uint32_t)(k % RING);
		uint32_t x, y;
		uint32_t spin = 0;
		for (;;) {
			x = ((volatile uint32_t*)sx)[slot * 64 + lane]; y = ((volatile uint32_t*)sy)[slot * 64 + lane];
			if (!__any((x == SENT) | (y == SENT))) { break; }
			if (++spin > (1u << 20)) { if (lane == 0) { atomicAdd(giveups, 1u); } break; }
			__builtin_amdgcn_s_sleep(1);
		}
		emit<MATH>(a, b, c, c0 + k, __uint_as_float(x), __uint_as_float(y));
	}
}

// the same ring, requests issued in BATCHES of B chunks (every B-th iteration asks for the B chunks AHEAD..AHEAD+B-1 later)
template<int RING, int B, int MATH>
__global__ __launch_bounds__(64) void k_ring_batched(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t chunks, unsigned int* giveups)
{
	__shared__ uint32_t sx[RING * 64];
	__shared__ uint32_t sy[RING * 64];
	const uint64_t per = chunks / gridDim.x; // multiple of 16
	const uint64_t c0 = (uint64_t)blockIdx.x * per;
	const int lane = threadIdx.x;
	const uint32_t bx = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)sx;
	const uint32_t by = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)sy;
	auto request_batch = [&](uint64_t k0) {
#pragma unroll
		for (int i = 0; i < B; ++i) { const uint32_t slot = (uint32_t)((k0 + i) % RING); sx[slot * 64 + lane] = SENT; sy[slot * 64 + lane] = SENT; }
		__builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
		for (int i = 0; i < B; ++i) {
			const uint32_t slot = (uint32_t)((k0 + i) % RING);
			const float2* src = in + (c0 + k0 + i) * 64 + lane;
			glds_dword(&src->x, __builtin_amdgcn_readfirstlane(bx + slot * 256));
			glds_dword(&src->y, __builtin_amdgcn_readfirstlane(by + slot * 256));
		}
	};
	request_batch(0);
	for (uint64_t k = 0; k < per; ++k) {
		if (k % B == 0 && k + B < per) { request_batch(k + B); } // RING >= 2 * B
		const uint32_t slot = (uint32_t)(k % RING);
		uint32_t x, y;
		uint32_t spin = 0;
		for (;;) {
			x = ((volatile uint32_t*)sx)[slot * 64 + lane]; y = ((volatile uint32_t*)sy)[slot * 64 + lane];
			if (!__any((x == SENT) | (y == SENT))) { break; }
			if (++spin > (1u << 20)) { if (lane == 0) { atomicAdd(giveups, 1u); } break; }
			__builtin_amdgcn_s_sleep(1);
		}
		emit<MATH>(a, b, c, c0 + k, __uint_as_float(x), __uint_as_float(y));
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

__global__ void k_check(const uint8_t* a, const float2* in, uint64_t chunks, unsigned int* bad)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks * 64; i += (uint64_t)gridDim.x * blockDim.x) {
		float x = in[i].x, y = in[i].y;
		for (int k = 0; k < 8; ++k) { x = x * 1.0001f + y; y = y * 0.9999f - x; }
		const uint32_t* q = (const uint32_t*)(a + i * 16);
		if (q[0] != __float_as_uint(x) || q[1] != __float_as_uint(y)) { atomicAdd(bad, 1u); }
	}
}

#define RUN(RING, AHEAD, g) { (void)hipMemset(ctr, 0, 8); float ms = best_ms([&] { hipLaunchKernelGGL((k_ring<RING, AHEAD, 8>), dim3(g), dim3(64), 0, 0, a, b, c, in, chunks, ctr); }); \
	hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, a, in, chunks, ctr + 1); unsigned int h[2]; (void)hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost); \
	printf("grid=%5d ring=%2d ahead=%2d: %.3f ms  write %.2f TB/s  (poll give-ups %u, wrong vertices %u)\n", g, RING, AHEAD, ms, (double)chunks * 2688 / ms / 1e9, h[0], h[1]); fflush(stdout); }

#define RUNB(RING, B, g) { (void)hipMemset(ctr, 0, 8); float ms = best_ms([&] { hipLaunchKernelGGL((k_ring_batched<RING, B, 8>), dim3(g), dim3(64), 0, 0, a, b, c, in, chunks, ctr); }); \
	hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, a, in, chunks, ctr + 1); unsigned int h[2]; (void)hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost); \
	printf("grid=%5d ring=%2d batch=%2d: %.3f ms  write %.2f TB/s  (poll give-ups %u, wrong vertices %u)\n", g, RING, B, ms, (double)chunks * 2688 / ms / 1e9, h[0], h[1]); fflush(stdout); }

int main()
{
	const uint64_t bytes = 6ull << 30;
	uint8_t* buf; float2* in; unsigned int* ctr;
	const uint64_t chunks = (bytes / 2688) / (32768 * 16) * (32768 * 16);
	if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess || hipMalloc(&in, chunks * 512 + 4096) != hipSuccess || hipMalloc(&ctr, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
	{ // distinct data so that the check means something
		float2* h = (float2*)malloc(chunks * 512);
		for (uint64_t i = 0; i < chunks * 64; ++i) { h[i].x = (float)(i % 9973) * 0.25f; h[i].y = (float)(i % 7919) * 0.5f; }
		(void)hipMemcpy(in, h, chunks * 512, hipMemcpyHostToDevice); free(h);
	}
	uint8_t* a = buf; uint8_t* b = buf + chunks * 1024 + 4096; uint8_t* c = buf + chunks * 1536 + 8192;
	for (int g : { 8192, 32768 }) {
		RUN(4, 1, g); RUN(8, 4, g); RUN(16, 12, g);
		RUNB(4, 2, g); RUNB(8, 4, g); RUNB(16, 8, g); RUNB(32, 16, g);
	}
	return 0;
}
