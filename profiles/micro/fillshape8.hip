// Micro-benchmark (tuning aid, round 3): why do k_fill's stores run at ~3 TB/s when the same three streams (16 + 8 + 18 B
// per lane) written densely run at 5.5 TB/s? Candidates tested here, stores only + 256 VALU per chunk, no reads:
//   dense        the reference point (fillshape.hip)
//   gaps G/P     the streams have HOLES: of every P chunks-worth of output only the first G are written (k_fill writes the
//                fill meshes of a draw, the stroke meshes in between are written later by k_stroke) -- in units of elements:
//                runs of `run` elements written, then `skip` elements skipped, per stream
//   last3        one lane in `every` writes 6 B of indices instead of 18 (the last corner of a mesh), leaving a 12 B hole
//   misalign     stream bases offset by 8 / 4 / 2 bytes (streams are only element aligned)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct V16 { uint32_t v[4]; };
struct V8 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) I9 { uint32_t a, b, c, d; uint16_t e; };
struct __attribute__((packed, aligned(2))) I3 { uint32_t a; uint16_t b; };

// element e of the flat WRITTEN stream lands at output element slot(e) = (e / run) * (run + skip) + e % run
template<int MATH>
__global__ __launch_bounds__(64) void k_gaps(uint8_t* a, uint8_t* b, uint8_t* c, uint64_t iters, uint32_t run, uint32_t skip, uint32_t every)
{
	const uint64_t perWave = iters / gridDim.x;
	const uint64_t e0 = (uint64_t)blockIdx.x * perWave * 64;
	for (uint64_t it = 0; it < perWave; ++it) {
		const uint64_t e = e0 + it * 64 + threadIdx.x;
		const uint64_t slot = skip ? (e / run) * (uint64_t)(run + skip) + e % run : e;
		float x = (float)threadIdx.x, y = (float)it;
#pragma unroll
		for (int k = 0; k < MATH; ++k) { x = x * 1.0001f + y; y = y * 0.9999f - x; }
		V16 q; q.v[0] = __float_as_uint(x); q.v[1] = __float_as_uint(y); q.v[2] = q.v[0] ^ 1; q.v[3] = q.v[1] ^ 1;
		*(V16*)(a + slot * 16) = q;
		V8 r; r.v[0] = q.v[0]; r.v[1] = q.v[1];
		*(V8*)(b + slot * 8) = r;
		if (every && (e % every) == every - 1) {
			I3 s; s.a = q.v[0]; s.b = (uint16_t)threadIdx.x;
			*(I3*)(c + slot * 18) = s;
		} else {
			I9 s; s.a = q.v[0]; s.b = q.v[1]; s.c = q.v[2]; s.d = q.v[3]; s.e = (uint16_t)threadIdx.x;
			*(I9*)(c + slot * 18) = s;
		}
	}
}

static void run(const char* name, uint8_t* buf, uint64_t cap, uint64_t iters, uint32_t runE, uint32_t skip, uint32_t every, uint32_t mis)
{
	const int g = 32768;
	const uint64_t slots = skip ? (iters * 64 / runE + 1) * (uint64_t)(runE + skip) : iters * 64;
	uint8_t* a = buf + (mis ? 8 : 0);
	uint8_t* b = buf + ((slots * 16 + 4096) & ~255ull) + (mis ? 4 : 0);
	uint8_t* c = buf + ((slots * 24 + 8192) & ~255ull) + (mis ? 2 : 0);
	if (slots * 42 + 16384 > cap) { printf("%s: does not fit\n", name); return; }
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep) {
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL((k_gaps<64>), dim3(g), dim3(64), 0, 0, a, b, c, iters, runE, skip, every);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) { best = ms; }
	}
	printf("%-44s %.3f ms  write %.2f TB/s\n", name, best, (double)iters * 2688 / best / 1e9);
}

int main()
{
	const uint64_t cap = 14ull << 30;
	uint8_t* buf;
	if (hipMalloc(&buf, cap) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void)hipMemset(buf, 0, cap);
	const int g = 32768;
	const uint64_t iters = ((6ull << 30) / 2688) / g * g;
	run("dense", buf, cap, iters, 0, 0, 0, 0);
	run("dense, misaligned bases", buf, cap, iters, 0, 0, 0, 1);
	run("last3 every 41", buf, cap, iters, 0, 0, 41, 0);
	run("runs of 41, skip 12 (tiger-like gaps)", buf, cap, iters, 41, 12, 0, 0);
	run("runs of 41, skip 12, last3, misaligned", buf, cap, iters, 41, 12, 41, 1);
	run("runs of 58, skip 28", buf, cap, iters, 58, 28, 58, 1);
	run("runs of 4096, skip 4096", buf, cap, iters, 4096, 4096, 0, 0);
	run("runs of 41, skip 1", buf, cap, iters, 41, 1, 0, 0);
	return 0;
}
