// Micro-benchmark (tuning aid): the store pattern of k_stroke's staged writer -- every lane writes 32 B of positions
// as two 16-B stores, 16 B of colours as two 8-B stores and 36 B of indices as three 12-B stores, so one store
// instruction covers every other 16-B piece of a 2 KB range -- against the same bytes written lane-contiguously, both
// with the 8 B/lane read stream of the real kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct V16 { uint32_t v[4]; };
struct V8 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) I6 { uint32_t a, b, c; };

template<bool STRIDED>
__global__ __launch_bounds__(64) void k_shape(uint8_t* a, uint8_t* b, uint8_t* c, const float2* in, uint64_t iters)
{
	const uint64_t perWave = iters / gridDim.x;
	uint8_t* pa = a + (uint64_t)blockIdx.x * perWave * 2048; // 32 B per lane
	uint8_t* pb = b + (uint64_t)blockIdx.x * perWave * 1024; // 16 B per lane
	uint8_t* pc = c + (uint64_t)blockIdx.x * perWave * 2304; // 36 B per lane
	const float2* pin = in + (uint64_t)blockIdx.x * perWave * 64;
	const int l = threadIdx.x;
	for (uint64_t it = 0; it < perWave; ++it) {
		const float2 p = pin[it * 64 + l];
		V16 q; q.v[0] = __float_as_uint(p.x); q.v[1] = __float_as_uint(p.y); q.v[2] = q.v[0] ^ 1; q.v[3] = q.v[1] ^ 1;
		V8 r; r.v[0] = q.v[0]; r.v[1] = q.v[1];
		I6 s; s.a = q.v[0]; s.b = q.v[1]; s.c = q.v[2];
		if (STRIDED) { // the real kernel: lane-major layout, one instruction = 64 pieces at the lane stride
			*(V16*)(pa + it * 2048 + l * 32) = q;
			*(V16*)(pa + it * 2048 + l * 32 + 16) = q;
			*(V8*)(pb + it * 1024 + l * 16) = r;
			*(V8*)(pb + it * 1024 + l * 16 + 8) = r;
			*(I6*)(pc + it * 2304 + l * 36) = s;
			*(I6*)(pc + it * 2304 + l * 36 + 12) = s;
			*(I6*)(pc + it * 2304 + l * 36 + 24) = s;
		} else { // same bytes, every instruction lane-contiguous
			*(V16*)(pa + it * 2048 + l * 16) = q;
			*(V16*)(pa + it * 2048 + 1024 + l * 16) = q;
			*(V8*)(pb + it * 1024 + l * 8) = r;
			*(V8*)(pb + it * 1024 + 512 + l * 8) = r;
			*(I6*)(pc + it * 2304 + l * 12) = s;
			*(I6*)(pc + it * 2304 + 768 + l * 12) = s;
			*(I6*)(pc + it * 2304 + 1536 + l * 12) = s;
		}
	}
}

int main()
{
	const uint64_t bytes = 6ull << 30;
	const int g = 32768;
	const uint64_t iters = (bytes / 5376) / g * g;
	uint8_t* buf; float2* in;
	if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess || hipMalloc(&in, iters * 512) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMemset(in, 0, iters * 512);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int strided = 1; strided >= 0; --strided) {
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			hipEventRecord(e0);
			if (strided) { hipLaunchKernelGGL(k_shape<true>, dim3(g), dim3(64), 0, 0, buf, buf + iters * 2048 + 4096, buf + iters * 3072 + 8192, in, iters); }
			else { hipLaunchKernelGGL(k_shape<false>, dim3(g), dim3(64), 0, 0, buf, buf + iters * 2048 + 4096, buf + iters * 3072 + 8192, in, iters); }
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			if (ms < best) { best = ms; }
		}
		printf("%-28s %.3f ms  write %.2f TB/s (+ %.2f TB/s read)\n", strided ? "lane-strided (k_stroke)" : "lane-contiguous", best, (double)iters * 5376 / best / 1e9, (double)iters * 512 / best / 1e9);
	}
	return 0;
}
