// vgx_mscan.h -- device-wide inclusive scan over an arbitrary monoid (max / sum mixes), three passes like vgx_scan.h:
// reduce contiguous slices -> scan the slice totals in one workgroup -> rescan every slice with its carry. The length is
// known to the host here (the path-set builder scans over commands), the grid is fixed, no atomics, deterministic.
//
// M interface (a trivially copyable struct of 32-bit words):
//   static M identity();  static M combine(M left, M right);  static M shfl_up(M v, int d);
// OP interface (all __device__):
//   uint32_t size() const;  M load(uint32_t i) const;  void store(uint32_t i, M incl, M own) const;  void finish(M total) const;
#ifndef VGX_MSCAN_H
#define VGX_MSCAN_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define VGX_MSCAN_BLOCKS 512
#define VGX_MSCAN_THREADS 256

template<class M, int T>
__device__ __forceinline__ M mscan_block_incl(M v, M* s_wave, M* blockTotal)
{
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const M t = M::shfl_up(v, d);
		if (lane >= d) { v = M::combine(t, v); }
	}
	if (lane == 63) { s_wave[wave] = v; }
	__syncthreads();
	M base = M::identity();
	M tot = M::identity();
#pragma unroll
	for (int w = 0; w < T / 64; ++w) {
		const M t = s_wave[w];
		if (w < wave) { base = M::combine(base, t); }
		tot = M::combine(tot, t);
	}
	__syncthreads();
	*blockTotal = tot;
	return M::combine(base, v);
}

__device__ __forceinline__ void mscan_slice(uint32_t n, uint32_t* lo, uint32_t* hi)
{
	uint32_t per = (n + VGX_MSCAN_BLOCKS - 1) / VGX_MSCAN_BLOCKS;
	per = (per + VGX_MSCAN_THREADS - 1) / VGX_MSCAN_THREADS * VGX_MSCAN_THREADS;
	const uint64_t l = (uint64_t)per * blockIdx.x;
	const uint64_t h = l + per;
	*lo = l > n ? n : (uint32_t)l;
	*hi = h > n ? n : (uint32_t)h;
}

template<class M, class OP>
__global__ __launch_bounds__(VGX_MSCAN_THREADS) void k_mscan_reduce(OP op, M* partial)
{
	__shared__ M s_wave[VGX_MSCAN_THREADS / 64];
	uint32_t lo, hi;
	mscan_slice(op.size(), &lo, &hi);
	M acc = M::identity();
	// stride T keeps the loads coalesced; the monoids used here are commutative (max, +, or), so the order inside a slice's
	// reduction does not matter (a non-commutative one would have to reduce tile by tile like k_mscan_apply)
	for (uint32_t i = lo + threadIdx.x; i < hi; i += VGX_MSCAN_THREADS) {
		acc = M::combine(acc, op.load(i));
	}
	M tot;
	mscan_block_incl<M, VGX_MSCAN_THREADS>(acc, s_wave, &tot);
	if (threadIdx.x == 0) { partial[blockIdx.x] = tot; }
}

template<class M, class OP>
__global__ __launch_bounds__(VGX_MSCAN_BLOCKS) void k_mscan_partials(OP op, M* partial)
{
	__shared__ M s_wave[VGX_MSCAN_BLOCKS / 64];
	__shared__ M s_all[VGX_MSCAN_BLOCKS];
	const M v = partial[threadIdx.x];
	M tot;
	const M incl = mscan_block_incl<M, VGX_MSCAN_BLOCKS>(v, s_wave, &tot);
	s_all[threadIdx.x] = incl;
	__syncthreads();
	partial[threadIdx.x] = threadIdx.x ? s_all[threadIdx.x - 1] : M::identity(); // exclusive: the carry into the slice
	if (threadIdx.x == 0) { op.finish(tot); }
}

template<class M, class OP>
__global__ __launch_bounds__(VGX_MSCAN_THREADS) void k_mscan_apply(OP op, const M* partial)
{
	__shared__ M s_wave[VGX_MSCAN_THREADS / 64];
	uint32_t lo, hi;
	mscan_slice(op.size(), &lo, &hi);
	if (lo >= hi) { return; }
	M carry = partial[blockIdx.x];
	for (uint32_t base = lo; base < hi; base += VGX_MSCAN_THREADS) {
		const uint32_t i = base + threadIdx.x;
		const M v = (i < hi) ? op.load(i) : M::identity();
		M tot;
		const M incl = mscan_block_incl<M, VGX_MSCAN_THREADS>(v, s_wave, &tot);
		if (i < hi) { op.store(i, M::combine(carry, incl), v); }
		carry = M::combine(carry, tot);
	}
}

template<class M, class OP>
static inline void vgx_monoid_scan(const OP& op, M* partial /* [VGX_MSCAN_BLOCKS] device */, hipStream_t s)
{
	hipLaunchKernelGGL((k_mscan_reduce<M, OP>), dim3(VGX_MSCAN_BLOCKS), dim3(VGX_MSCAN_THREADS), 0, s, op, partial);
	hipLaunchKernelGGL((k_mscan_partials<M, OP>), dim3(1), dim3(VGX_MSCAN_BLOCKS), 0, s, op, partial);
	hipLaunchKernelGGL((k_mscan_apply<M, OP>), dim3(VGX_MSCAN_BLOCKS), dim3(VGX_MSCAN_THREADS), 0, s, op, partial);
}

#endif
