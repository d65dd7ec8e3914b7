// Micro-benchmark (tuning aid), follow-up of fillshape2: the 8 B/lane read stream costs ~0.6 ms beside 6 GB of stores
// whatever the wave -> address mapping. Does the WIDTH / cache policy of the read or of the stores matter?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct V16 { uint32_t v[4]; };
struct V8 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) I9 { uint32_t a, b, c, d; uint16_t e; };
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template<int MATH, bool NT>
__device__ __forceinline__ void emit(uint8_t* a, uint8_t* b, uint8_t* c, uint64_t ch, float x, float y)
{
#pragma unroll
	for (int k = 0; k < MATH; ++k) { x = x * 1.0001f + y; y = y * 0.9999f - x; }
	const uint32_t q0 = __float_as_uint(x), q1 = __float_as_uint(y);
	if (NT) {
		u4 q = { q0, q1, q0 ^ 1, q1 ^ 1 };
		__builtin_nontemporal_store(q, (u4*)(a + ch * 1024 + threadIdx.x * 16));
		u2 r = { q0, q1 };
		__builtin_nontemporal_store(r, (u2*)(b + ch * 512 + threadIdx.x * 8));
	} else {
		V16 q; q.v[0] = q0; q.v[1] = q1; q.v[2] = q0 ^ 1; q.v[3] = q1 ^ 1;
		*(V16*)(a + ch * 1024 + threadIdx.x * 16) = q;
		V8 r; r.v[0] = q0; r.v[1] = q1;
		*(V8*)(b + ch * 512 + threadIdx.x * 8) = r;
	}
	I9 s; s.a = q0; s.b = q1; s.c = q0 ^ 1; s.d = q1 ^ 1; s.e = (uint16_t)threadIdx.x;
	*(I9*)(c + ch * 1152 + threadIdx.x * 18) = s;
}

// MODE 0: 8 B/lane read per chunk. 1: one 16 B/lane read per TWO chunks, redistributed with shuffles. 2: one 16 B/lane
// read (lanes 0..31 only: 32 lanes x 16 B = the chunk's 512 B) per chunk. 3: nontemporal 8 B loads. 4: no read.
template<int MODE, int MATH, bool NTS>
__global__ __launch_bounds__(64) void k_var(uint8_t* a, uint8_t* b, uint8_t* c, const float* in, uint64_t chunks)
{
	const uint64_t per = chunks / gridDim.x; // even
	const uint64_t c0 = (uint64_t)blockIdx.x * per;
	const int lane = threadIdx.x;
	if (MODE == 1) {
		for (uint64_t i = 0; i < per; i += 2) {
			const float4 v = *(const float4*)(in + (c0 + i) * 128 + lane * 4); // elements 2*lane, 2*lane+1 of the 128-element pair
			// chunk A element e (lane e): owner lane e/2, component e%2; chunk B element e: owner lane 32 + e/2
			const int srcA = lane >> 1, srcB = 32 + (lane >> 1);
			const bool odd = lane & 1;
			const float ax = __shfl(odd ? v.z : v.x, srcA), ay = __shfl(odd ? v.w : v.y, srcA);
			// NOTE: select must happen on the SOURCE lane by the DESTINATION's parity: do both and pick
			const float x0 = __shfl(v.x, srcA), y0 = __shfl(v.y, srcA), x1 = __shfl(v.z, srcA), y1 = __shfl(v.w, srcA);
			const float X0 = __shfl(v.x, srcB), Y0 = __shfl(v.y, srcB), X1 = __shfl(v.z, srcB), Y1 = __shfl(v.w, srcB);
			(void)ax; (void)ay;
			emit<MATH, NTS>(a, b, c, c0 + i, odd ? x1 : x0, odd ? y1 : y0);
			emit<MATH, NTS>(a, b, c, c0 + i + 1, odd ? X1 : X0, odd ? Y1 : Y0);
		}
	} else {
		for (uint64_t i = 0; i < per; ++i) {
			float x = (float)lane, y = (float)i;
			if (MODE == 0) { const float2 p = *(const float2*)(in + (c0 + i) * 128 + lane * 2); x = p.x; y = p.y; }
			if (MODE == 3) { const u2 p = __builtin_nontemporal_load((const u2*)(in + (c0 + i) * 128 + lane * 2)); x = __uint_as_float(p.x); y = __uint_as_float(p.y); }
			if (MODE == 2) {
				float4 v = make_float4(0, 0, 0, 0);
				if (lane < 32) { v = *(const float4*)(in + (c0 + i) * 128 + lane * 4); }
				const int src = lane >> 1;
				const float x0 = __shfl(v.x, src), y0 = __shfl(v.y, src), x1 = __shfl(v.z, src), y1 = __shfl(v.w, src);
				x = (lane & 1) ? x1 : x0; y = (lane & 1) ? y1 : y0;
			}
			emit<MATH, NTS>(a, b, c, c0 + i, x, y);
		}
	}
}

// pure read: 8 or 16 B per lane, summed (kept alive through a never-true store)
template<int W>
__global__ __launch_bounds__(64) void k_read(const float* in, uint64_t chunks, float* sink)
{
	const uint64_t per = chunks / gridDim.x;
	const uint64_t c0 = (uint64_t)blockIdx.x * per;
	float acc = 0.0f;
	if (W == 8) { for (uint64_t i = 0; i < per; ++i) { const float2 p = *(const float2*)(in + (c0 + i) * 128 + threadIdx.x * 2); acc += p.x + p.y; } }
	else { for (uint64_t i = 0; i < per; i += 2) { const float4 p = *(const float4*)(in + (c0 + i) * 128 + threadIdx.x * 4); acc += p.x + p.y + p.z + p.w; } }
	if (acc == 1234.5f) { *sink = acc; }
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

#define RUN(MODE, NTS, label) { float ms = best_ms([&] { hipLaunchKernelGGL((k_var<MODE, 64, NTS>), dim3(g), dim3(64), 0, 0, a, b, c, in, chunks); }); \
	printf("%-58s %.3f ms  write %.2f TB/s\n", label, ms, (double)chunks * 2688 / ms / 1e9); }

int main()
{
	const uint64_t bytes = 6ull << 30;
	uint8_t* buf; float* in;
	const int g = 32768;
	const uint64_t chunks = (bytes / 2688) / (32768 * 16) * (32768 * 16);
	if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess || hipMalloc(&in, chunks * 512 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void)hipMemset(in, 0, chunks * 512);
	uint8_t* a = buf; uint8_t* b = buf + chunks * 1024 + 4096; uint8_t* c = buf + chunks * 1536 + 8192;
	{
		float m8 = best_ms([&] { hipLaunchKernelGGL((k_read<8>), dim3(g), dim3(64), 0, 0, in, chunks, (float*)buf); });
		float m16 = best_ms([&] { hipLaunchKernelGGL((k_read<16>), dim3(g), dim3(64), 0, 0, in, chunks, (float*)buf); });
		printf("pure read %.2f GB: 8 B/lane %.3f ms %.2f TB/s | 16 B/lane %.3f ms %.2f TB/s\n", chunks * 512 / 1e9, m8, chunks * 512 / m8 / 1e9, m16, chunks * 512 / m16 / 1e9);
	}
	RUN(4, false, "stores only");
	RUN(4, true, "stores only, pos/colour nontemporal");
	RUN(0, false, "8 B/lane read per chunk");
	RUN(0, true, "8 B/lane read per chunk, nontemporal pos/colour stores");
	RUN(3, false, "nontemporal 8 B/lane read per chunk");
	RUN(3, true, "nontemporal read + nontemporal stores");
	RUN(1, false, "16 B/lane read per two chunks + shuffles");
	RUN(2, false, "16 B/lane read by 32 lanes per chunk + shuffles");
	return 0;
}
