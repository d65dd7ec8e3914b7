// vgx_scan.h -- device-wide exclusive prefix sums whose length lives in DEVICE memory.
//
// The batch pipeline needs four scans (command instances per draw, polyline/sub-path/mesh counts per
// draw, stroker elements per mesh, vertices/indices per mesh). Their lengths are results of earlier
// kernels, so the launch geometry must not depend on them: a fixed grid of VGX_SCAN_BLOCKS blocks, each
// owning one contiguous slice of the input, three passes: reduce slices -> scan the slice totals in one
// block -> rescan each slice with its carry. No host round trip, no atomics, deterministic.
//
// OP interface (all __device__):
//   uint64_t size() const;                 // number of items (read from device memory)
//   Sum3     load(uint64_t i) const;       // up to four uint64 fields per item
//   void     store(uint64_t i, Sum3 excl); // exclusive prefix for item i
//   void     finish(Sum3 total);           // called once by one thread with the grand total
//   uint64_t apply_size() const;           // OPTIONAL: only the prefixes of items [0, apply_size()) are needed (<= size());
//                                          // evaluated in the third pass, i.e. it may depend on what load() found in the first
#ifndef VGX_SCAN_H
#define VGX_SCAN_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

#define VGX_SCAN_BLOCKS 512
#define VGX_SCAN_THREADS 256

struct Sum3 // (historic name) four independent uint64 sums carried through one scan
{
	uint64_t a, b, c, d;
};

__device__ __forceinline__ Sum3 sum3_add(Sum3 x, Sum3 y)
{
	Sum3 r;
	r.a = x.a + y.a; r.b = x.b + y.b; r.c = x.c + y.c; r.d = x.d + y.d;
	return r;
}
__device__ __forceinline__ Sum3 sum3_zero() { Sum3 r; r.a = 0; r.b = 0; r.c = 0; r.d = 0; return r; }

__device__ __forceinline__ Sum3 sum3_shfl_up(Sum3 v, int d)
{
	Sum3 r;
	r.a = __shfl_up((unsigned long long)v.a, d);
	r.b = __shfl_up((unsigned long long)v.b, d);
	r.c = __shfl_up((unsigned long long)v.c, d);
	r.d = __shfl_up((unsigned long long)v.d, d);
	return r;
}

// Inclusive scan across the block (T threads, T/64 waves). Returns the inclusive value; *blockTotal
// receives the sum over the block. s_wave must hold T/64 entries.
template<int T>
__device__ __forceinline__ Sum3 block_incl_scan(Sum3 v, Sum3* s_wave, Sum3* blockTotal)
{
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const Sum3 t = sum3_shfl_up(v, d);
		if (lane >= d) { v = sum3_add(v, t); }
	}
	if (lane == 63) { s_wave[wave] = v; }
	__syncthreads();
	Sum3 base = sum3_zero();
	Sum3 tot = sum3_zero();
#pragma unroll
	for (int w = 0; w < T / 64; ++w) {
		const Sum3 t = s_wave[w];
		if (w < wave) { base = sum3_add(base, t); }
		tot = sum3_add(tot, t);
	}
	__syncthreads();
	*blockTotal = tot;
	return sum3_add(v, base);
}

__device__ __forceinline__ void scan_slice(uint64_t n, uint64_t* lo, uint64_t* hi)
{
	uint64_t per = (n + VGX_SCAN_BLOCKS - 1) / VGX_SCAN_BLOCKS;
	per = (per + VGX_SCAN_THREADS - 1) / VGX_SCAN_THREADS * VGX_SCAN_THREADS;
	uint64_t l = per * blockIdx.x;
	uint64_t h = l + per;
	if (l > n) { l = n; }
	if (h > n) { h = n; }
	*lo = l;
	*hi = h;
}

template<class OP>
__global__ __launch_bounds__(VGX_SCAN_THREADS) void k_scan_reduce(OP op, Sum3* partial)
{
	__shared__ Sum3 s_wave[VGX_SCAN_THREADS / 64];
	uint64_t lo, hi;
	scan_slice(op.size(), &lo, &hi);
	Sum3 acc = sum3_zero();
	for (uint64_t i = lo + threadIdx.x; i < hi; i += VGX_SCAN_THREADS) {
		acc = sum3_add(acc, op.load(i));
	}
	Sum3 tot;
	block_incl_scan<VGX_SCAN_THREADS>(acc, s_wave, &tot);
	if (threadIdx.x == 0) { partial[blockIdx.x] = tot; }
}

template<class OP>
__global__ __launch_bounds__(VGX_SCAN_BLOCKS) void k_scan_partials(OP op, Sum3* partial)
{
	__shared__ Sum3 s_wave[VGX_SCAN_BLOCKS / 64];
	const Sum3 v = partial[threadIdx.x];
	Sum3 tot;
	const Sum3 incl = block_incl_scan<VGX_SCAN_BLOCKS>(v, s_wave, &tot);
	Sum3 excl;
	excl.a = incl.a - v.a; excl.b = incl.b - v.b; excl.c = incl.c - v.c; excl.d = incl.d - v.d;
	partial[threadIdx.x] = excl;
	if (threadIdx.x == 0) { op.finish(tot); }
}

template<class OP, class = void>
struct ScanApplyLimit { static __device__ __forceinline__ uint64_t get(const OP&, uint64_t n) { return n; } };
template<class OP>
struct ScanApplyLimit<OP, std::void_t<decltype(std::declval<const OP&>().apply_size())>>
{
	static __device__ __forceinline__ uint64_t get(const OP& op, uint64_t n) { const uint64_t a = op.apply_size(); return a < n ? a : n; }
};

template<class OP>
__global__ __launch_bounds__(VGX_SCAN_THREADS) void k_scan_apply(OP op, const Sum3* partial)
{
	__shared__ Sum3 s_wave[VGX_SCAN_THREADS / 64];
	uint64_t lo, hi;
	const uint64_t n = op.size();
	scan_slice(n, &lo, &hi);
	const uint64_t lim = ScanApplyLimit<OP>::get(op, n); // block-uniform
	if (lo >= lim) { return; }
	if (hi > lim) { hi = lim; }
	Sum3 carry = partial[blockIdx.x];
	for (uint64_t base = lo; base < hi; base += VGX_SCAN_THREADS) {
		const uint64_t i = base + threadIdx.x;
		const Sum3 v = (i < hi) ? op.load(i) : sum3_zero();
		Sum3 tot;
		const Sum3 incl = block_incl_scan<VGX_SCAN_THREADS>(v, s_wave, &tot);
		if (i < hi) {
			Sum3 e;
			e.a = carry.a + incl.a - v.a; e.b = carry.b + incl.b - v.b; e.c = carry.c + incl.c - v.c; e.d = carry.d + incl.d - v.d;
			op.store(i, e);
		}
		carry = sum3_add(carry, tot);
	}
}

// Small inputs: the whole scan in ONE workgroup (a frame-sized batch is launch-latency bound: three dependent
// launches per scan x five scans cost ~8 us of a 127 us single-drawing call; beyond one tile the three-pass form is faster).
#define VGX_SCAN_SINGLE_THREADS 1024
#define VGX_SCAN_SINGLE_MAX 1024
// The whole scan by ONE workgroup of T threads; callable from inside a larger single-workgroup kernel (every thread of
// the block must call it; s_wave holds T / 64 entries; ends with the block synchronised).
template<class OP, int T>
__device__ __forceinline__ void block_scan_all(const OP& op, Sum3* s_wave)
{
	const uint64_t n = op.size();
	Sum3 carry = sum3_zero();
	for (uint64_t base = 0; base < n; base += T) {
		const uint64_t i = base + threadIdx.x;
		const Sum3 v = (i < n) ? op.load(i) : sum3_zero();
		Sum3 tot;
		const Sum3 incl = block_incl_scan<T>(v, s_wave, &tot);
		if (i < n) {
			Sum3 e;
			e.a = carry.a + incl.a - v.a; e.b = carry.b + incl.b - v.b; e.c = carry.c + incl.c - v.c; e.d = carry.d + incl.d - v.d;
			op.store(i, e);
		}
		carry = sum3_add(carry, tot);
	}
	if (threadIdx.x == 0) { op.finish(carry); }
	__syncthreads();
}

template<class OP>
__global__ __launch_bounds__(VGX_SCAN_SINGLE_THREADS) void k_scan_single(OP op)
{
	__shared__ Sum3 s_wave[VGX_SCAN_SINGLE_THREADS / 64];
	block_scan_all<OP, VGX_SCAN_SINGLE_THREADS>(op, s_wave);
}

// maxItems: an upper bound of op.size() the HOST knows (the size itself lives in device memory); picks the launch shape.
template<class OP>
static inline void vgx_device_scan(const OP& op, Sum3* partial /* [VGX_SCAN_BLOCKS] device */, hipStream_t s, uint64_t maxItems = ~0ull)
{
	if (maxItems <= VGX_SCAN_SINGLE_MAX) {
		hipLaunchKernelGGL(k_scan_single<OP>, dim3(1), dim3(VGX_SCAN_SINGLE_THREADS), 0, s, op);
		return;
	}
	hipLaunchKernelGGL(k_scan_reduce<OP>, dim3(VGX_SCAN_BLOCKS), dim3(VGX_SCAN_THREADS), 0, s, op, partial);
	hipLaunchKernelGGL(k_scan_partials<OP>, dim3(1), dim3(VGX_SCAN_BLOCKS), 0, s, op, partial);
	hipLaunchKernelGGL(k_scan_apply<OP>, dim3(VGX_SCAN_BLOCKS), dim3(VGX_SCAN_THREADS), 0, s, op, partial);
}

#endif
