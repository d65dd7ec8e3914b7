// vgx_wave.h -- wavefront (64-lane) primitives for gfx950: ballot masks, segment heads, prefix scans.
// Device only. A "segment" is a run of consecutive lanes that belong to the same draw / sub-path /
// mesh; its head lane is flagged in a 64-bit ballot mask.
#ifndef VGX_WAVE_H
#define VGX_WAVE_H

#include <hip/hip_runtime.h>
#include <rocprim/warp/warp_scan.hpp>
#include <stdint.h>

#define VGX_WAVE 64

__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __ballot(p ? 1 : 0); }
__device__ __forceinline__ uint64_t lanemask_le(int lane) { return (2ull << lane) - 1ull; }      // bits 0..lane
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }      // bits 0..lane-1
__device__ __forceinline__ uint64_t lanemask_ge(int lane) { return ~((1ull << lane) - 1ull); }   // bits lane..63

// Highest flagged lane <= `lane`, or -1 when the segment started in an earlier chunk.
__device__ __forceinline__ int seg_head(uint64_t heads, int lane)
{
	const uint64_t m = heads & lanemask_le(lane);
	return m ? 63 - __clzll((long long)m) : -1;
}

// Lanes of my segment up to and including me (segment = [head, lane]); head < 0 means from lane 0.
__device__ __forceinline__ uint64_t seg_mask_upto(int head, int lane)
{
	return lanemask_le(lane) & (head < 0 ? ~0ull : lanemask_ge(head));
}

// Inclusive prefix sums over the wavefront: rocprim's wave64 scan lowers to DPP row_shr / row_bcast moves (no LDS
// crossbar round trips, unlike a ds_bpermute shuffle ladder).
__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
	(void)lane;
	using WS = rocprim::warp_scan<int, VGX_WAVE>;
	typename WS::storage_type st;
	int out;
	WS().inclusive_scan(v, out, st);
	return out;
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
	(void)lane;
	using WS = rocprim::warp_scan<uint32_t, VGX_WAVE>;
	typename WS::storage_type st;
	uint32_t out;
	WS().inclusive_scan(v, out, st);
	return out;
}

// Broadcast of lane `src`'s value where `src` is WAVE-UNIFORM: v_readlane, no LDS crossbar.
__device__ __forceinline__ int wave_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint32_t wave_bcast_u32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ uint64_t wave_bcast_u64(uint64_t v, int src)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
	return ((uint64_t)hi << 32) | lo;
}

// Neighbour lanes through DPP whole-wave shifts (gfx9 wave_shr:1 / wave_shl:1): lane l reads lane l-1 / l+1.
// Lane 0 (resp. 63) receives `edge`.
__device__ __forceinline__ float wave_from_prev(float v, float edge)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_from_next(float v, float edge)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t wave_from_prev_u32(uint32_t v, uint32_t edge)
{
	return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false);
}

// Value of `v` on lane `src` (src may differ per lane).
__device__ __forceinline__ int wave_read(int v, int src) { return __shfl(v, src); }
__device__ __forceinline__ uint32_t wave_read_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src); }

// Exclusive-scan value relative to the segment head: excl[lane] - excl[head], or carry + excl[lane]
// when the segment continues from the previous chunk.
__device__ __forceinline__ int seg_rel(int excl, int head, int carry)
{
	const int base = __shfl(excl, head < 0 ? 0 : head);
	return head < 0 ? carry + excl : excl - base;
}

// lower_bound over a device array of uint64 (first index i in [lo,hi) with a[i] >= key, else hi)
__device__ __forceinline__ uint64_t lower_bound_u64(const uint64_t* a, uint64_t lo, uint64_t hi, uint64_t key)
{
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (a[mid] < key) { lo = mid + 1; } else { hi = mid; }
	}
	return lo;
}

// upper_bound - 1: last index i in [lo,hi) with a[i] <= key (assumes a[lo] <= key)
__device__ __forceinline__ uint64_t find_owner_u64(const uint64_t* a, uint64_t lo, uint64_t hi, uint64_t key)
{
	while (hi - lo > 1) {
		const uint64_t mid = (lo + hi) >> 1;
		if (a[mid] <= key) { lo = mid; } else { hi = mid; }
	}
	return lo;
}

// ---- walking a sorted uint64 prefix array without per-segment binary searches ----------------------------
// A wave processes a CONTIGUOUS run of 64-item segments, so the owner range of segment s+1 starts where the
// range of segment s ended. advance_lower_bound moves forward with one coalesced 64-entry load per step.

// First index i in [start, n) with P[i] >= key, or n when there is none (P has n+1 sorted entries and P[n]
// is the grand total). Wave-uniform result.
__device__ __forceinline__ uint64_t advance_lower_bound(const uint64_t* P, uint64_t start, uint64_t n, uint64_t key, int lane)
{
	uint64_t m = start;
	for (;;) {
		const uint64_t idx = m + (uint64_t)lane;
		const uint64_t v = (idx < n) ? P[idx] : ~0ull;
		const int cnt = __popcll(wave_ballot(v < key)); // sorted: the lanes below the key form a prefix
		m += (uint64_t)cnt;
		if (cnt < VGX_WAVE) { return m; }
	}
}

// Items per segment. Draws (flatten) and meshes (stroke) are bucketed whole into segments of the flat item stream and
// a segment is walked in 64-item chunks, so the last chunk of every segment is partly empty: with ~25-item draws,
// 64-item segments run at ~67 % lane use, 512-item segments at ~94 %. Small batches keep small segments so that
// they still spread over many waves (>= 4 segments per wave before growing). Wave-uniform.
__device__ __forceinline__ uint64_t vgx_segment_items(uint64_t total, uint32_t grid)
{
	uint64_t per = total / ((uint64_t)grid * 4 * VGX_WAVE);
	per = per < 1 ? 1 : (per > 8 ? 8 : per);
	return per * VGX_WAVE;
}

// 64-entry window of the prefix array held one entry per lane: w = P[first + lane] (or ~0 past `last`).
// window_owner returns the offset k (0..63) of the LAST window entry <= key (requires W[0] <= key); *pv gets
// that entry. Six shuffle steps, no memory traffic.
__device__ __forceinline__ int window_owner(uint64_t w, uint64_t key, uint64_t* pv)
{
	int lo = 0;
	uint64_t vlo = __shfl((unsigned long long)w, 0);
#pragma unroll
	for (int step = 32; step >= 1; step >>= 1) {
		const int cand = lo + step;
		const uint64_t v = __shfl((unsigned long long)w, cand & 63);
		if (cand < VGX_WAVE && v <= key) { lo = cand; vlo = v; }
	}
	*pv = vlo;
	return lo;
}

// Same search on a window that was reduced to 32-bit offsets relative to the chunk start: rel[k] = clamp(W[k] - chunk,
// 0, 64) (entries at or before the chunk start become 0, entries past the chunk 64), key = lane. Half the shuffles.
__device__ __forceinline__ int window_owner_rel(uint32_t rel, uint32_t key)
{
	int lo = 0;
#pragma unroll
	for (int step = 32; step >= 1; step >>= 1) {
		const int cand = lo + step;
		const uint32_t v = (uint32_t)__shfl((int)rel, cand & 63);
		if (cand < VGX_WAVE && v <= key) { lo = cand; }
	}
	return lo;
}

__device__ __forceinline__ uint32_t window_rel(uint64_t prefix, uint64_t chunk)
{
	return prefix <= chunk ? 0u : (prefix - chunk > 64ull ? 64u : (uint32_t)(prefix - chunk));
}

#endif
