// vgx_flat1.h -- shared by vgx_flat1.hip (kernels) and vgx_api.hip (vgx_flatten): arguments of the ordered one-walk flatten.
#ifndef VGX_FLAT1_H
#define VGX_FLAT1_H

#include "vgx_internal.h"

// Look-back record of one segment: running sums of polyline vertices (w[0]) and of sub-paths / meshes (w[1], 31 bits each);
// bits 62-63 of every word = 0 nothing yet / 1 the segment's own totals / 2 inclusive prefix. 16 bytes: a wave reads 64 of them
// with one 1 KB request.
struct __attribute__((aligned(16))) VgxF1Seg { unsigned long long w[2]; };

struct VgxF1Args
{
	uint64_t* seg_draw;   // [segments + 1] first draw of every segment
	VgxF1Seg* segs;       // [segments] look-back records of the tickets (cleared by the segment-table kernel)
	VgxF1Seg* grps;       // [segments / 64 + 1] ... of the groups of 64 tickets
	uint64_t cap_poly;    // the caller's capacities (vertices / sub-path records)
	uint64_t cap_subs;
	int pass;             // 0: the normal run; 1: the second run of a batch in which the first one found degenerate draws
	int read_flags;       // dinfo[d].flags may already mark serial draws
	int has_empty;        // the path set has paths without commands: their draws' records are written by the neighbours
	unsigned long long tag; // identifies the batch (path set generation, draw count) in the host's copy of the totals
	uint32_t seg_max;     // command instances per segment bucket at most (64; less for batches whose chunks would overflow the leaf list)
};

// Command instances per segment bucket. Draws are bucketed whole by their FIRST command instance, so a segment is longer than
// its bucket by what its last draw hangs over: buckets of 64 minus the mean draw length keep most segments inside ONE 64-command
// chunk (one walk); paths of one or two commands (a million moveTo + cubicTo pairs) fill the chunk exactly.
__host__ __device__ inline uint64_t vgx_f1_segment_items(uint64_t totalCmds, uint64_t ndraws, uint32_t segMax)
{
	const uint64_t avg = totalCmds / (ndraws ? ndraws : 1);
	uint64_t s = avg <= 2 ? 64 : (avg >= 32 ? 32 : 64 - avg);
	if (segMax >= 2 && s > segMax) { s = segMax; }
	return s;
}

// Private-memory pending stack for the exact serial builder (k_f1_serial_count_list only; k_flat1 itself has no scratch).
struct PrivStackF1
{
	float s[VGX_CUBIC_MAX_PENDING * 6];
	__device__ __forceinline__ void push(int level, float ax, float ay, float bx, float by, float cx, float cy)
	{
		float* p = s + level * 6;
		p[0] = ax; p[1] = ay; p[2] = bx; p[3] = by; p[4] = cx; p[5] = cy;
	}
	__device__ __forceinline__ void pop(int level, float& ax, float& ay, float& bx, float& by, float& cx, float& cy)
	{
		const float* p = s + level * 6;
		ax = p[0]; ay = p[1]; bx = p[2]; by = p[3]; cx = p[4]; cy = p[5];
	}
};

void vgx_launch_flat1(const VgxFlattenArgs& a, const VgxF1Args& x, int waves, int cap, bool hasStaticSerial, hipStream_t s);
void vgx_launch_flat1_publish(const VgxFlattenArgs& a, const VgxF1Args& x, vgx_sizes* devSizes, uint32_t* devStatus, hipStream_t s);
void vgx_launch_flatten_serial(bool emit, const VgxFlattenArgs& a, hipStream_t s); // k_flatten_serial alone (vgx_flatten.hip)

#endif
