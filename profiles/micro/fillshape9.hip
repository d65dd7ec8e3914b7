// Micro-benchmark (tuning aid, round 3): the output streams of a frame are written by TWO kernels, each leaving holes for the
// other (k_fill: runs of ~41 elements, then a hole of ~12 slots that k_stroke fills later). fillshape8 measured what the
// holes cost ONE kernel. Question here: what does the PAIR cost, and would it help if one wave wrote a region's runs and,
// right after them, the same region's holes (so that the partially written lines at the run / hole boundaries are completed
// while they still sit in the CU's L2)?
//   two kernels      k_part<runs> over everything, then k_part<holes> over everything (what ships)
//   one kernel, K    every wave: the runs of a region of K periods, then that region's holes; K = 4 .. 256
//   dense            every slot written once, in order (the bound)
// Streams as in fillshape8: 16 + 8 + 18 B per slot, stores only + 64 VALU per chunk.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct V16 { uint32_t v[4]; };
struct V8 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) I9 { uint32_t a, b, c, d; uint16_t e; };

__device__ __forceinline__ void put(uint8_t* a, uint8_t* b, uint8_t* c, uint64_t slot, uint32_t tag)
{
	float x = (float)tag, y = (float)(slot & 1023);
#pragma unroll
	for (int k = 0; k < 64; ++k) { x = x * 1.0001f + y; y = y * 0.9999f - x; }
	V16 q; q.v[0] = __float_as_uint(x); q.v[1] = __float_as_uint(y); q.v[2] = q.v[0] ^ 1; q.v[3] = q.v[1] ^ 1;
	*(V16*)(a + slot * 16) = q;
	V8 r; r.v[0] = q.v[0]; r.v[1] = q.v[1];
	*(V8*)(b + slot * 8) = r;
	I9 s; s.a = q.v[0]; s.b = q.v[1]; s.c = q.v[2]; s.d = q.v[3]; s.e = (uint16_t)tag;
	*(I9*)(c + slot * 18) = s;
}

// part 0: the runs (first `run` slots of every period), part 1: the holes (the other `skip` slots); elements of a part are
// packed into 64-lane chunks. periods [p0, p1) of the wave.
__device__ __forceinline__ void write_part(uint8_t* a, uint8_t* b, uint8_t* c, uint64_t p0, uint64_t p1, uint32_t run, uint32_t skip, int part)
{
	const uint32_t len = part ? skip : run, off = part ? run : 0u;
	const uint64_t n = (p1 - p0) * len;
	for (uint64_t e = threadIdx.x; e < n; e += 64) {
		const uint64_t p = p0 + e / len;
		put(a, b, c, p * (run + skip) + off + e % len, (uint32_t)threadIdx.x);
	}
}

__global__ __launch_bounds__(64) void k_part(uint8_t* a, uint8_t* b, uint8_t* c, uint64_t periods, uint32_t run, uint32_t skip, int part)
{
	const uint64_t per = (periods + gridDim.x - 1) / gridDim.x;
	const uint64_t p0 = (uint64_t)blockIdx.x * per, p1 = p0 + per < periods ? p0 + per : periods;
	if (p0 < p1) { write_part(a, b, c, p0, p1, run, skip, part); }
}

__global__ __launch_bounds__(64) void k_regions(uint8_t* a, uint8_t* b, uint8_t* c, uint64_t periods, uint32_t run, uint32_t skip, uint32_t K)
{
	const uint64_t per = (periods + gridDim.x - 1) / gridDim.x;
	const uint64_t p0 = (uint64_t)blockIdx.x * per, p1 = p0 + per < periods ? p0 + per : periods;
	for (uint64_t r = p0; r < p1; r += K) {
		const uint64_t r1 = r + K < p1 ? r + K : p1;
		write_part(a, b, c, r, r1, run, skip, 0);
		write_part(a, b, c, r, r1, run, skip, 1);
	}
}

__global__ __launch_bounds__(64) void k_dense(uint8_t* a, uint8_t* b, uint8_t* c, uint64_t slots)
{
	const uint64_t per = ((slots + gridDim.x - 1) / gridDim.x + 63) / 64 * 64;
	const uint64_t s0 = (uint64_t)blockIdx.x * per, s1 = s0 + per < slots ? s0 + per : slots;
	for (uint64_t s = s0 + threadIdx.x; s < s1; s += 64) { put(a, b, c, s, (uint32_t)threadIdx.x); }
}

int main()
{
	const uint64_t cap = 9ull << 30;
	uint8_t* buf;
	if (hipMalloc(&buf, cap) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void)hipMemset(buf, 0, cap);
	const uint32_t run = 41, skip = 12;
	const uint64_t periods = (8ull << 30) / 42 / (run + skip);
	const uint64_t slots = periods * (run + skip);
	uint8_t* a = buf; uint8_t* b = buf + ((slots * 16 + 4096) & ~255ull); uint8_t* c = buf + ((slots * 24 + 8192) & ~255ull);
	const int g = 32768;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const double bytes = (double)slots * 42;
	for (int mode = 0; mode < 9; ++mode) {
		float best = 1e9f;
		const uint32_t K = mode >= 2 ? (1u << (mode - 1)) : 0; // 2, 4, 8, ... 128
		for (int rep = 0; rep < 3; ++rep) {
			(void)hipEventRecord(e0);
			if (mode == 0) { hipLaunchKernelGGL(k_dense, dim3(g), dim3(64), 0, 0, a, b, c, slots); }
			else if (mode == 1) {
				hipLaunchKernelGGL(k_part, dim3(g), dim3(64), 0, 0, a, b, c, periods, run, skip, 0);
				hipLaunchKernelGGL(k_part, dim3(g), dim3(64), 0, 0, a, b, c, periods, run, skip, 1);
			} else { hipLaunchKernelGGL(k_regions, dim3(g), dim3(64), 0, 0, a, b, c, periods, run, skip, K); }
			(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
			float ms; (void)hipEventElapsedTime(&ms, e0, e1);
			if (ms < best) { best = ms; }
		}
		if (mode == 0) { printf("dense                       %.3f ms  %.2f TB/s\n", best, bytes / best / 1e9); }
		else if (mode == 1) { printf("two kernels (runs, holes)   %.3f ms  %.2f TB/s\n", best, bytes / best / 1e9); }
		else { printf("one kernel, regions of %3u  %.3f ms  %.2f TB/s\n", K, best, bytes / best / 1e9); }
	}
	return 0;
}
