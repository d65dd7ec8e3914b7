// Micro-benchmark (tuning aid): HBM write / read+write ceilings for the access shapes of k_fill / k_stroke:
// persistent one-wave workgroups, each owning a contiguous range, 64 lanes x W bytes per store instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template<int W> struct Vec { uint32_t v[W / 4]; };

template<int W>
__global__ __launch_bounds__(64) void k_write(uint8_t* out, uint64_t bytes)
{
	const uint64_t perWave = (bytes / gridDim.x) / (64 * W) * (64 * W);
	uint8_t* p = out + (uint64_t)blockIdx.x * perWave;
	Vec<W> x;
	for (int i = 0; i < W / 4; ++i) { x.v[i] = threadIdx.x + i; }
	for (uint64_t o = 0; o < perWave; o += 64 * W) {
		*(Vec<W>*)(p + o + threadIdx.x * W) = x;
	}
}

// three streams like k_fill: 16 B, 8 B, 18 B (as 16 + 2) per lane per iteration
__global__ __launch_bounds__(64) void k_write3(uint8_t* a, uint8_t* b, uint8_t* c, uint64_t iters, int mis)
{
	const uint64_t perWave = iters / gridDim.x;
	uint8_t* pa = a + (uint64_t)blockIdx.x * perWave * 1024 + mis * 24;
	uint8_t* pb = b + (uint64_t)blockIdx.x * perWave * 512 + mis * 12;
	uint8_t* pc = c + (uint64_t)blockIdx.x * perWave * 1152 + mis * 54;
	Vec<16> x; Vec<8> y;
	for (int i = 0; i < 4; ++i) { x.v[i] = threadIdx.x + i; }
	y.v[0] = 1; y.v[1] = 2;
	for (uint64_t it = 0; it < perWave; ++it) {
		*(Vec<16>*)(pa + it * 1024 + threadIdx.x * 16) = x;
		*(Vec<8>*)(pb + it * 512 + threadIdx.x * 8) = y;
		struct __attribute__((packed, aligned(2))) I9 { uint32_t a, b, c, d; uint16_t e; } q;
		q.a = 1; q.b = 2; q.c = 3; q.d = threadIdx.x; q.e = 7;
		*(I9*)(pc + it * 1152 + threadIdx.x * 18) = q;
	}
}

int main()
{
	const uint64_t bytes = 6ull << 30;
	uint8_t* buf;
	if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int grids[] = { 4096, 32768 };
	for (int g : grids) {
		for (int w : { 4, 8, 16 }) {
			float best = 1e9f;
			for (int rep = 0; rep < 4; ++rep) {
				hipEventRecord(e0);
				if (w == 4) { hipLaunchKernelGGL(k_write<4>, dim3(g), dim3(64), 0, 0, buf, bytes); }
				if (w == 8) { hipLaunchKernelGGL(k_write<8>, dim3(g), dim3(64), 0, 0, buf, bytes); }
				if (w == 16) { hipLaunchKernelGGL(k_write<16>, dim3(g), dim3(64), 0, 0, buf, bytes); }
				hipEventRecord(e1); hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1);
				if (ms < best) { best = ms; }
			}
			printf("write grid=%d %2d B/lane: %.3f ms  %.2f TB/s\n", g, w, best, (double)bytes / best / 1e9);
		}
		const uint64_t iters = (bytes / (1024 + 512 + 1152)) / g * g;
		for (int mis = 0; mis < 2; ++mis) {
			float best = 1e9f;
			for (int rep = 0; rep < 4; ++rep) {
				hipEventRecord(e0);
				hipLaunchKernelGGL(k_write3, dim3(g), dim3(64), 0, 0, buf, buf + iters * 1024 + 4096, buf + iters * 1536 + 8192, iters, mis);
				hipEventRecord(e1); hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1);
				if (ms < best) { best = ms; }
			}
			printf("write3 (16+8+18 B/lane, 3 streams, misaligned=%d) grid=%d: %.3f ms  %.2f TB/s\n", mis, g, best, (double)iters * 2688 / best / 1e9);
		}
	}
	return 0;
}
