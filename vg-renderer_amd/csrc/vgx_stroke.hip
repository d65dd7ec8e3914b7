// vgx_stroke.hip -- batch stroke / convex-fill / AA-fringe mesh generation on gfx950
// (replaces vg::strokerXXX, reference src/stroker.cpp).
//
// Work decomposition
//   One lane = one ELEMENT = one polyline vertex of one mesh: a polygon corner of a convex fill, or a cap /
//   join of a polyline stroke. Fills and strokes run in two kernels (k_fill, k_stroke), each over its own flat
//   element stream (exclusive scan of the polyline length over the meshes of that class); a wavefront owns a
//   contiguous run of 64-element segments and therefore whole meshes, walks the prefix array cooperatively
//   (vgx_wave.h) and finds every lane's mesh with six shuffles.
//   The reference builds each mesh sequentially (running m_NumVertices / m_NumIndices and the prevSegment*ID
//   bookkeeping, stroker.cpp:1401-1410). Here every element
//     A. gets its neighbours' vertices and the previous segment's direction from the adjacent lanes (one vertex
//        load and one vec2Dir per element) and computes its own vertex / index counts (data dependent only for
//        Round joins / caps),
//     B. gets its vertex / index base inside the mesh from a wave prefix scan segmented by mesh with a carry across
//        chunks (strokes) or in closed form (fills),
//     C. gets the previous element's exit rail IDs (prevSegment{LeftAA,Left,Right,RightAA}ID) from the
//        neighbouring lane (shuffle, carry across chunks), and
//     D. writes its vertices, colours and uint16 indices straight to their final place.
//   k_mesh_prepare (one lane per mesh) precomputes the per-mesh constants (half widths, fill orientation, colour);
//   k_round_sizes (one wave per mesh) sizes the meshes with Round joins -- every other mesh size is closed-form and
//   was written together with the mesh descriptor.
//
// Every emitted position / colour / index follows the cited reference lines; the rails formulation is the one of
// SURVEY.md appendix B.
#include "vgx_internal.h"
#include "vgx_wave.h"
#include "vgx_elem.h"
#include "vgx_fastmath.h"

namespace {

// One lane per mesh: per-mesh constants for the element kernels. (Meshes with Round joins -- the only data-dependent
// sizes -- are sized by k_round_sizes, one wave per mesh; every other mesh was sized in closed form by flatten.)
__global__ __launch_bounds__(256) void k_mesh_prepare(VgxStrokeArgs A)
{
	if (A.totals->status != VGX_OK) {
		return;
	}
	const uint64_t numMeshes = A.totals->sizes.num_meshes;
	for (uint64_t mi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; mi < numMeshes; mi += (uint64_t)gridDim.x * blockDim.x) {
		const VgxMeshDesc md = A.mdesc[mi];
		const vgx_draw* dr = A.draws + md.draw;
		const VgxMeshPrep pr = mesh_prep(md, dr, A.poly);
		A.mprep[mi] = pr;
	}
}

// Sizes of meshes with Round joins, the only ones whose vertex / index counts depend on the geometry (numArcPoints per
// join, stroker.cpp:1146, 1592): one wave per such mesh, lanes stride over its elements and sum what k_stroke will
// emit for each. Needs no element prefix, so the scan over meshes can produce element and vertex / index offsets
// together afterwards. Returns at once when the batch has no Round joins.
__global__ __launch_bounds__(VGX_WAVE) void k_round_sizes(VgxStrokeArgs A)
{
	if (A.totals->status != VGX_OK || A.totals->num_round_meshes == 0) {
		return;
	}
	const int lane = threadIdx.x;
	const uint64_t numMeshes = A.totals->sizes.num_meshes;
	for (uint64_t mi = blockIdx.x; mi < numMeshes; mi += gridDim.x) {
		if (A.mtab[mi].num_vertices != VGX_MESH_NEEDS_COUNT) { // wave-uniform
			continue;
		}
		const VgxMeshDesc md = A.mdesc[mi];
		const VgxMeshPrep pr = A.mprep[mi];
		uint32_t sv, si;
		round_mesh_size(make_mesh_ctx(md, pr, A.draws, 0, A.poly), lane, &sv, &si);
		if (lane == 0) {
			A.mtab[mi].num_vertices = sv;
			A.mtab[mi].num_indices = si;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// k_fill: strokerConvexFill / strokerConvexFillAA (stroker.cpp:334-365, 713-807)
//   FILL_AA element j: vertices 2j (inner, colour c) and 2j+1 (outer, colour c0) and the 9 (last element: 3)
//   index positions [9j, 9j+9) of the mesh, whose values are closed-form in (N, position).
//   FILL element j: vertex j (= polyline vertex) and fan triangle (0, j+1, j+2).
// ------------------------------------------------------------------------------------------------
// Unaligned wide stores: the output streams are only element-aligned (8 / 4 / 2 bytes), gfx950 global stores
// handle that natively (unaligned access mode), so every lane issues ONE dwordx4 for its two positions, ONE
// dwordx2 for its two colours and ONE dwordx4 + ONE short for its nine indices -- no alignment-dependent
// divergence.
// Per-mesh record held one per lane for a window of 64 consecutive meshes (refilled every few dozen chunks):
// the element lanes fetch their mesh's fields with shuffles instead of a dependent chain of global loads.
struct __attribute__((aligned(16))) FillRec // 48 bytes, one per mesh of the window, in LDS
{
	uint64_t polyFirst, firstV, firstI;
	uint32_t N, kind, color;
	float aa;
	uint32_t ibase;    // assembly: vertices in front of the mesh inside its vertex buffer (0 when not armed)
	uint32_t pad;
};

struct FillWindow
{
	uint64_t prefix;   // elem_prefix[wbase + lane] (or ~0 past the end), read back from LDS: no VMEM-pending register
	                   // lives across the pipelined loop (the compiler would guard it with s_waitcnt vmcnt(0))
	FillRec* rec;      // s_win
	uint64_t* pre;     // s_pre
};

// Loads the window of 64 mesh records starting at wbase into LDS (one mesh per lane) and waits for it.
__device__ __forceinline__ void fill_window_load(const VgxStrokeArgs& A, FillWindow& W, uint64_t wbase, uint64_t numMeshes, int lane)
{
	const uint64_t idx = wbase + (uint64_t)lane;
	const uint64_t prefix = (idx <= numMeshes) ? A.elem_prefix[idx] : ~0ull;
	FillRec r;
	r.polyFirst = 0; r.firstV = 0; r.firstI = 0; r.N = 3; r.kind = VGX_MESH_FILL; r.color = 0; r.aa = 0.0f; r.ibase = 0; r.pad = 0;
	if (idx < numMeshes) {
		const VgxMeshDesc md = A.mdesc[idx];
		const VgxMeshPrep pr = A.mprep[idx];
		r.polyFirst = md.poly_first; r.N = md.poly_n; r.kind = VGX_MD_KIND(md.kind);
		r.color = pr.color; r.aa = pr.f0;
		r.firstV = A.mtab[idx].first_vertex;
		r.firstI = A.mtab[idx].first_index;
		if (A.mesh_base) { r.ibase = A.mesh_base[idx]; }
	}
	__syncthreads(); // one-wave workgroup: lanes may still be reading the previous window
	W.rec[lane] = r;
	W.pre[lane] = prefix;
	__syncthreads();
	W.prefix = W.pre[lane];
}

// ---- the walk of k_fill: runs of chunks through an LDS ring ----------------------------------------------------------
// What bounds this kernel is how its 8 B / element read stream mixes with its 42 B / element store streams in the memory
// system, not its own waits: chunk-by-chunk interleaving (one 512 B read, then 2.7 KB of stores, per wave) runs the
// access mix at ~3.3 TB/s of writes whatever the prefetch depth (even with the reads decoupled from vmcnt altogether,
// profiles/micro/fillshape6.hip), a wave that requests the vertices of 8-16 chunks back to back and then emits them
// reaches 4.2-4.3 TB/s (profiles/micro/fillshape4.hip, fillshape6.hip). The real kernel gains far less from it (2-5 %:
// with its loads issued but never waited for it runs no faster, with its stores removed it takes 1.1 ms instead of 2.0,
// i.e. it is the traffic itself, not a wait) but the structure is also the simplest: a wave processes its elements in RUNS of
// VGX_FILL_RUN chunks: (1) owner search and vertex address of every chunk of the run, all vertex loads issued back to
// back, (2) the run's vertices parked in an LDS ring, (3) the chunks emitted one after the other. A corner's two
// neighbours come from the adjacent lanes (DPP) or, at chunk edges and where a polygon wraps around, from the ring --
// not from extra loads; only the four vertices just outside the run (the first mesh's vertex 0 and the vertex in front
// of the run, the last mesh's last vertex and the vertex behind the run) are loaded with the burst.
#ifndef VGX_FILL_RUN
#define VGX_FILL_RUN 8 /* measured on Tiger x10k, same box: 4: 2.15 ms, 8: 2.02-2.09, 12: 2.07, 16: 2.07-2.13 (146 VGPRs) */
#endif
#define VGX_FILL_RING (VGX_FILL_RUN * VGX_WAVE)

struct FillRunState // what the emit phase of a run needs besides the ring
{
	int k[VGX_FILL_RUN];        // owner mesh = window entry (valid lanes)
	uint32_t j[VGX_FILL_RUN];   // element index inside the mesh; 0xFFFFFFFF = lane has no element
};

// Owner search of one chunk (window covers it): window entry and element index of every lane.
__device__ __forceinline__ void fill_owner(const FillWindow& W, uint64_t chunk, uint64_t elemEnd, int lane, int* kOut, uint32_t* jOut)
{
	const uint64_t ei = chunk + (uint64_t)lane;
	const bool valid = ei < elemEnd;
	const uint32_t wrel = window_rel(W.prefix, chunk);
	const int k = window_owner_rel(wrel, valid ? (uint32_t)lane : 0u);
	const uint32_t orel = (uint32_t)__shfl((int)wrel, k);
	const int firstOwner = __popcll(wave_ballot(W.prefix <= chunk)) - 1;
	const uint64_t headBase = wave_bcast_u64(W.prefix, firstOwner < 0 ? 0 : firstOwner); // mesh that owns the chunk's first element
	*kOut = k;
	*jOut = valid ? (orel > 0 ? (uint32_t)lane - orel : (uint32_t)(ei - headBase)) : 0xFFFFFFFFu;
}

// Emits chunk i of the run [run0, run0 + runLen). ring[r] = vertex of the run's element r, ring[VGX_FILL_RING + 0..3] =
// the four outside vertices {in front of the run, first mesh's vertex 0, behind the run, last mesh's last vertex}.
__device__ __forceinline__ void fill_emit_ring(const VgxStrokeArgs& A, const FillWindow& W, const float2* ring, int i, int k, uint32_t jj, uint64_t run0, uint32_t runLen, uint64_t elemEnd, int lane)
{
	FillFetch F;
	F.valid = jj != 0xFFFFFFFFu;
	F.j = F.valid ? jj : 0u;
	const FillRec r = W.rec[k];
	F.N = r.N;
	F.color = r.color; F.aa = r.aa; F.firstV = r.firstV; F.firstI = r.firstI; F.ibase = r.ibase; F.mi = 0;
	F.aaElem = F.valid && r.kind == VGX_MESH_FILL_AA;
	const uint32_t e = (uint32_t)i * VGX_WAVE + (uint32_t)lane; // my element, relative to the run
	const uint64_t ei = run0 + e;
	F.prevInWave = lane > 0 && F.j > 0;
	F.nextInWave = lane < VGX_WAVE - 1 && F.j + 1 < F.N && ei + 1 < elemEnd;
	const float2 me = ring[F.valid ? e : 0u];
	F.p1 = v2(me.x, me.y);
	F.pNextB = F.p1; F.pPrevB = F.p1;
	if (F.aaElem && !F.nextInWave) {
		uint32_t slot;
		if (F.j + 1 < F.N) { slot = e + 1 < runLen ? e + 1 : VGX_FILL_RING + 2; }       // next corner: in the run, or the vertex behind it
		else { slot = e >= F.j ? e - F.j : VGX_FILL_RING + 1; }                           // wrap to the mesh's vertex 0: in the run, or the first mesh's
		const float2 q = ring[slot];
		F.pNextB = v2(q.x, q.y);
	}
	if (F.aaElem && !F.prevInWave) {
		uint32_t slot;
		if (F.j > 0) { slot = e > 0 ? e - 1 : VGX_FILL_RING + 0; }                         // previous corner: in the run, or the vertex in front of it
		else { slot = e + (F.N - 1) < runLen ? e + (F.N - 1) : VGX_FILL_RING + 3; }       // wrap to the mesh's last vertex: in the run, or the last mesh's
		const float2 q = ring[slot];
		F.pPrevB = v2(q.x, q.y);
	}
	fill_emit_chunk(A.pos, A.color, A.idx, F);
}

// A chunk in which more than 63 mesh records begin (zero-length entries of stroke-only sub-paths between fills): every
// lane searches its mesh in memory. Rare.
__device__ __forceinline__ uint64_t fill_chunk_slow(const VgxStrokeArgs& A, uint64_t chunk, uint64_t elemEnd, uint64_t mlo, uint64_t numMeshes, int lane)
{
	FillFetch F;
	const uint64_t ei = chunk + (uint64_t)lane;
	const bool valid = ei < elemEnd;
	F.valid = valid; F.j = 0; F.N = 3; F.color = 0; F.aa = 0.0f; F.firstV = 0; F.firstI = 0; F.ibase = 0; F.mi = mlo;
	F.aaElem = false; F.prevInWave = false; F.nextInWave = false;
	F.p1 = v2(0.0f, 0.0f); F.pNextB = F.p1; F.pPrevB = F.p1;
	if (valid) {
		const uint64_t mi = find_owner_u64(A.elem_prefix, mlo, numMeshes, ei);
		const VgxMeshDesc md = A.mdesc[mi];
		const VgxMeshPrep pr = A.mprep[mi];
		F.mi = mi;
		F.j = (uint32_t)(ei - A.elem_prefix[mi]);
		F.N = md.poly_n; F.color = pr.color; F.aa = pr.f0;
		F.firstV = A.mtab[mi].first_vertex; F.firstI = A.mtab[mi].first_index;
		F.ibase = A.mesh_base ? A.mesh_base[mi] : 0u;
		F.aaElem = VGX_MD_KIND(md.kind) == VGX_MESH_FILL_AA;
		const float* vtx = A.poly + 2 * md.poly_first;
		F.p1 = ldv(vtx, F.j);
		if (F.aaElem) { // no neighbour shortcuts here: both loads, always
			F.pNextB = ldv(vtx, F.j + 1 < F.N ? F.j + 1 : 0);
			F.pPrevB = ldv(vtx, F.j > 0 ? F.j - 1 : F.N - 1);
		}
	}
	fill_emit_chunk(A.pos, A.color, A.idx, F);
	const int nvalid = (int)((elemEnd - chunk) < (uint64_t)VGX_WAVE ? (elemEnd - chunk) : (uint64_t)VGX_WAVE);
	return wave_bcast_u64(F.mi, nvalid - 1);
}

#ifndef VGX_FILL_OCC
#define VGX_FILL_OCC
#endif
__global__ __launch_bounds__(VGX_WAVE) VGX_FILL_OCC void k_fill(VgxStrokeArgs A)
{
	__shared__ FillRec s_win[VGX_WAVE];
	__shared__ uint64_t s_pre[VGX_WAVE];
	__shared__ float2 s_ring[VGX_FILL_RING + 4];
	const int lane = threadIdx.x;
	if (A.totals->status != VGX_OK) {
		return;
	}
	const uint64_t numMeshes = A.totals->sizes.num_meshes;
	const uint64_t totalElems = A.elem_prefix[numMeshes];
	const uint64_t numSegments = (totalElems + VGX_WAVE - 1) / VGX_WAVE;
	const uint64_t segsPerWave = (numSegments + gridDim.x - 1) / gridDim.x;
	const uint64_t seg0 = (uint64_t)blockIdx.x * segsPerWave;
	const uint64_t seg1 = (seg0 + segsPerWave < numSegments) ? seg0 + segsPerWave : numSegments;
	if (seg0 >= seg1) {
		return;
	}
	// This wave's elements are the contiguous range [pos, elemEnd); whole meshes are NOT required here (a fill element
	// only needs its own mesh record and its two neighbours), so the walk is a plain 64-element stride.
	uint64_t pos = seg0 * VGX_WAVE;
	const uint64_t elemEnd = (seg1 * VGX_WAVE < totalElems) ? seg1 * VGX_WAVE : totalElems;
	uint64_t mcur = find_owner_u64(A.elem_prefix, 0, numMeshes, pos); // last mesh with prefix <= pos
	uint64_t wbase = mcur;
	FillWindow W;
	W.rec = s_win; W.pre = s_pre;
	fill_window_load(A, W, wbase, numMeshes, lane);

	while (pos < elemEnd) { // wave-uniform
		uint64_t wlast = wave_bcast_u64(W.prefix, VGX_WAVE - 1);
		if (!(wlast > pos + (VGX_WAVE - 1))) { // the window does not cover the next chunk
			wbase = mcur;
			fill_window_load(A, W, wbase, numMeshes, lane);
			wlast = wave_bcast_u64(W.prefix, VGX_WAVE - 1);
			if (!(wlast > pos + (VGX_WAVE - 1))) { // more than 63 mesh records inside one chunk
				mcur = fill_chunk_slow(A, pos, elemEnd, wbase, numMeshes, lane);
				pos += VGX_WAVE;
				continue;
			}
		}
		const uint64_t covered = (wlast - pos) >> 6;                    // chunks the window covers from pos on
		const uint64_t left = (elemEnd - pos + (VGX_WAVE - 1)) >> 6;
		uint64_t nn = covered < left ? covered : left;
		nn = nn < (uint64_t)VGX_FILL_RUN ? nn : (uint64_t)VGX_FILL_RUN;
		const int n = (int)nn;
		const uint64_t run0 = pos;
		const uint64_t runEnd = run0 + nn * VGX_WAVE < elemEnd ? run0 + nn * VGX_WAVE : elemEnd;
		const uint32_t runLen = (uint32_t)(runEnd - run0);

		// (1) owner search + vertex request of every chunk of the run
		FillRunState R;
		float2 v[VGX_FILL_RUN];
		// heap INDEX of every chunk's vertex, not a pointer: the selected pointers lost their address space on the way
		// through the unrolled loop and the eight vertex loads of a run became flat_load (which also counts on lgkmcnt,
		// so every LDS wait of the emit phase waited for the heap as well)
		uint64_t src[VGX_FILL_RUN];
		const float2* const heap = (const float2*)A.poly;
#pragma unroll
		for (int i = 0; i < VGX_FILL_RUN; ++i) {
			R.k[i] = 0; R.j[i] = 0xFFFFFFFFu; v[i] = make_float2(0.0f, 0.0f); src[i] = 0;
			if (i < n) { // wave-uniform
				fill_owner(W, run0 + (uint64_t)i * VGX_WAVE, elemEnd, lane, &R.k[i], &R.j[i]);
				const uint64_t polyFirst = W.rec[R.k[i]].polyFirst;
				if (R.j[i] != 0xFFFFFFFFu) { src[i] = polyFirst + R.j[i]; }
#ifdef VGX_EXP_WRAPREAD /* tuning experiment: the polyline read stream wraps inside 2 MB (L2 resident); results are wrong */
				src[i] &= 0x3FFFFull;
#endif
			}
		}
		// all vertex loads of the run back to back (every lane loads: lanes without an element re-read the heap's first vertex)
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int i = 0; i < VGX_FILL_RUN; ++i) {
			if (i < n) { v[i] = heap[src[i]]; }
		}
		__builtin_amdgcn_sched_barrier(0);
		// the four vertices just outside the run, lanes 0..3: {in front of the run, first mesh's vertex 0, behind the run,
		// last mesh's last vertex}; the first / last element's mesh and index come from lane 0 of chunk 0 / the last valid lane
		float2 edge = make_float2(0.0f, 0.0f);
		{
			const int lastChunkValid = (int)(runLen - (uint32_t)(n - 1) * VGX_WAVE); // valid lanes of the run's last chunk, >= 1
			int kF = wave_bcast(R.k[0], 0), kL = 0;
			uint32_t jF = wave_bcast_u32(R.j[0], 0), jL = 0;
#pragma unroll
			for (int i = 0; i < VGX_FILL_RUN; ++i) {
				if (i == n - 1) { kL = wave_bcast(R.k[i], lastChunkValid - 1); jL = wave_bcast_u32(R.j[i], lastChunkValid - 1); }
			}
			if (lane < 4) {
				const FillRec rr = W.rec[lane < 2 ? kF : kL];
				uint32_t idx;
				if (lane == 0) { idx = jF > 0 ? jF - 1 : 0; }
				else if (lane == 1) { idx = 0; }
				else if (lane == 2) { idx = jL + 1 < rr.N ? jL + 1 : 0; }
				else { idx = rr.N - 1; }
#ifdef VGX_EXP_WRAPREAD
				edge = *(const float2*)(A.poly + 2 * ((rr.polyFirst + idx) & 0x3FFFFull));
#else
				edge = *(const float2*)(A.poly + 2 * (rr.polyFirst + idx));
#endif
			}
			mcur = wbase + (uint64_t)kL; // owner of the run's last element: where the next window (if one is needed) starts
		}
		// (2) park the run in the ring
		__syncthreads(); // one-wave workgroup: lanes may still be reading the previous run
#pragma unroll
		for (int i = 0; i < VGX_FILL_RUN; ++i) {
			if (i < n) { s_ring[i * VGX_WAVE + lane] = v[i]; }
		}
		if (lane < 4) { s_ring[VGX_FILL_RING + lane] = edge; }
		__syncthreads();
		// (3) emit
#pragma unroll
		for (int i = 0; i < VGX_FILL_RUN; ++i) {
			if (i < n) { fill_emit_ring(A, W, s_ring, i, R.k[i], R.j[i], run0, runLen, elemEnd, lane); }
		}
		pos = run0 + nn * VGX_WAVE;
	}
}

// ------------------------------------------------------------------------------------------------
// k_fill2: the same meshes as k_fill with a third of the instructions (round 3).
// k_fill is bound by its own instruction stream as much as by memory (VALU busy 1.1 of its 1.9 ms; with the polyline reads
// made L2 hits it only gains 0.2 ms, profiles/README.md): ~250 VALU per 64-element chunk, of which owner search 40, the
// compiler's generic IEEE 1/x and sqrt sequences 80 (three divisions + two square roots: every chunk has lanes on both
// sides of each branch), ring addressing, DPP neighbour exchange with its hazard nops, 64-bit addressing. k_fill2:
//   - every lane loads its OWN three vertices (previous / own / next with the polygon's wrap-around): the two extra loads
//     hit L1 (the neighbours' lines), and there is no ring, no DPP shift, no boundary case and no branch in the element;
//   - 1 / sqrt(lenSqr) and 1 / cross through vgx_fastmath.h (correctly rounded over their whole domain, checked exhaustively
//     on the device; out-of-domain values -- coordinates beyond 1e15 -- take the generic sequence under a branch that never runs);
//   - owner search = one LDS histogram of the mesh heads inside the chunk + one wave prefix sum (3 LDS operations in a row
//     instead of a ladder of 6 dependent ds_bpermute);
//   - 32-byte mesh records with 32-bit offsets relative to the window's first mesh; output addresses = wave-uniform 64-bit
//     base + 32-bit offset.
// Non-AA fills and chunks the window cannot cover go through fill_chunk_slow (everything from memory), as in k_fill.
// Meshes whose polyline lies beyond heap index 2^32 (a > 32 GB heap) send their window through fill_chunk_slow as well.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) Fill2Rec // 32 bytes, one per mesh of the window, in LDS
{
	uint32_t poly;      // heap index of the mesh's vertex 0
	uint32_t N;
	uint32_t firstV;    // relative to the window's first mesh
	uint32_t firstI;
	uint32_t color;
	float aa;
	uint32_t ibase;
	uint32_t kind;
};

__device__ __forceinline__ V2 v2dir_fast(V2 a, V2 b) // vec2Dir, stroker.cpp:31-38
{
	const float dx = b.x - a.x;
	const float dy = b.y - a.y;
	const float lenSqr = dx * dx + dy * dy;
	float inv = vgx_rsqrt_rn(lenSqr);
	if (__builtin_expect(!(lenSqr <= 0x1p100f), 0)) { inv = vgm_rsqrt(lenSqr); } // outside the checked domain (incl. NaN / Inf)
	const float invLen = lenSqr < VGM_EPSILON ? 0.0f : inv;
	return v2(dx * invLen, dy * invLen);
}

__device__ __forceinline__ V2 v2extrude_fast(V2 d01, V2 d12) // calcExtrusionVector, stroker.cpp:40-53
{
	V2 v = v2ccw(d01);
	const float c = v2cross(d12, d01);
	const float ac = vgm_abs(c);
	if (ac > (1.0f / 100.0f)) {
		float r = vgx_rcp_rn(c);
		if (__builtin_expect(!(ac <= 0x1p100f), 0)) { r = 1.0f / c; }
		v = v2mul(v2sub(d01, d12), r);
	}
	return v;
}

template<int RUN>
__global__ __launch_bounds__(VGX_WAVE) void k_fill2(VgxStrokeArgs A)
{
	__shared__ Fill2Rec s_win[VGX_WAVE];
	__shared__ uint32_t s_cnt[RUN][VGX_WAVE];
	const int lane = threadIdx.x;
	if (A.totals->status != VGX_OK) {
		return;
	}
	const uint64_t numMeshes = A.totals->sizes.num_meshes;
	const uint64_t totalElems = A.elem_prefix[numMeshes];
	const uint64_t numSegments = (totalElems + VGX_WAVE - 1) / VGX_WAVE;
	const uint64_t segsPerWave = (numSegments + gridDim.x - 1) / gridDim.x;
	const uint64_t seg0 = (uint64_t)blockIdx.x * segsPerWave;
	const uint64_t seg1 = (seg0 + segsPerWave < numSegments) ? seg0 + segsPerWave : numSegments;
	if (seg0 >= seg1) {
		return;
	}
	uint64_t pos = seg0 * VGX_WAVE;
	const uint64_t elemEnd = (seg1 * VGX_WAVE < totalElems) ? seg1 * VGX_WAVE : totalElems;
#pragma unroll
	for (int c = 0; c < RUN; ++c) { s_cnt[c][lane] = 0; }
	uint64_t mcur = find_owner_u64(A.elem_prefix, 0, numMeshes, pos); // last mesh with prefix <= pos
	// window state: wave-uniform bases + one 32-bit relative prefix per lane
	uint64_t wbase = 0, winE0 = 0, winV0 = 0, winI0 = 0;
	uint32_t wrel = 0;          // elem_prefix[wbase + lane] - winE0 (0x7FFFFFFF past the table)
	uint32_t wlast = 0;         // lane 63's
	bool haveWindow = false;
	const float2* const heap = (const float2*)A.poly;

	while (pos < elemEnd) { // wave-uniform
		if (!haveWindow || !(wlast > (uint32_t)(pos - winE0) + (VGX_WAVE - 1))) { // the window does not cover the next chunk
			wbase = mcur;
			const uint64_t idx = wbase + (uint64_t)lane;
			const uint64_t prefix = (idx <= numMeshes) ? A.elem_prefix[idx] : ~0ull;
			Fill2Rec r;
			r.poly = 0; r.N = 3; r.firstV = 0; r.firstI = 0; r.color = 0; r.aa = 0.0f; r.ibase = 0; r.kind = VGX_MESH_FILL;
			uint64_t fv = 0, fi = 0;
			bool far = false;
			if (idx < numMeshes) {
				const VgxMeshDesc md = A.mdesc[idx];
				const VgxMeshPrep pr = A.mprep[idx];
				far = ((md.poly_first + md.poly_n) >> 32) != 0;
				r.poly = (uint32_t)md.poly_first; r.N = md.poly_n; r.kind = VGX_MD_KIND(md.kind);
				r.color = pr.color; r.aa = pr.f0;
				fv = A.mtab[idx].first_vertex;
				fi = A.mtab[idx].first_index;
				if (A.mesh_base) { r.ibase = A.mesh_base[idx]; }
			}
			winE0 = wave_bcast_u64(prefix, 0);
			winV0 = wave_bcast_u64(fv, 0);
			winI0 = wave_bcast_u64(fi, 0);
			r.firstV = (uint32_t)(fv - winV0); r.firstI = (uint32_t)(fi - winI0);
			const uint64_t d = prefix - winE0;
			wrel = (prefix == ~0ull || d > 0x7FFFFFFFull) ? 0x7FFFFFFFu : (uint32_t)d;
			wlast = wave_bcast_u32(wrel, VGX_WAVE - 1);
			__syncthreads(); // one-wave workgroup: lanes may still be reading the previous window
			s_win[lane] = r;
			__syncthreads();
			haveWindow = true;
			if (wave_ballot(far) != 0 || !(wlast > (uint32_t)(pos - winE0) + (VGX_WAVE - 1))) { // 32-bit heap indices do not reach, or more than 63 mesh records inside one chunk
				mcur = fill_chunk_slow(A, pos, elemEnd, wbase, numMeshes, lane);
				pos += VGX_WAVE;
				haveWindow = false;
				continue;
			}
		}
		const uint32_t p0 = (uint32_t)(pos - winE0);
		const uint64_t covered = (uint64_t)((wlast - p0) >> 6);          // chunks the window covers from pos on
		const uint64_t left = (elemEnd - pos + (VGX_WAVE - 1)) >> 6;
		uint64_t nn = covered < left ? covered : left;
		nn = nn < (uint64_t)RUN ? nn : (uint64_t)RUN;
		const int n = (int)nn;

		// (A) owner of every element of the run, its record, its three vertex requests
		int kk[RUN];
		uint32_t jj[RUN];
		Fill2Rec rec[RUN];
		float2 vP[RUN], v1[RUN], vN[RUN];
		bool val[RUN];
		uint64_t slowMask = 0; // chunks with a non-AA element (wave-uniform bit per chunk)
#pragma unroll
		for (int c = 0; c < RUN; ++c) {
			kk[c] = 0; jj[c] = 0; val[c] = false; rec[c] = s_win[0];
			vP[c] = make_float2(0.0f, 0.0f); v1[c] = vP[c]; vN[c] = vP[c];
			if (c < n) { // wave-uniform
				const uint32_t pc = p0 + (uint32_t)c * VGX_WAVE;
				const int rel = (int)(wrel - pc);              // where my window entry begins relative to the chunk (<= 0: before / at its start)
				const int count0 = __popcll(wave_ballot(rel <= 0));
				if (rel > 0 && rel < VGX_WAVE) { atomicAdd(&s_cnt[c][rel], 1u); }
				__builtin_amdgcn_wave_barrier();
				const uint32_t heads = s_cnt[c][lane];
				s_cnt[c][lane] = 0;
				const int k = count0 - 1 + (int)wave_incl_scan_u32(heads, lane);
				const int relk = __shfl(rel, k);
				const bool valid = pos + (uint64_t)c * VGX_WAVE + (uint64_t)lane < elemEnd;
				const Fill2Rec r = s_win[k];
				const uint32_t j = valid ? (uint32_t)(lane - relk) : 0u;
				kk[c] = k; jj[c] = j; val[c] = valid; rec[c] = r;
				if (wave_ballot(valid && r.kind != VGX_MESH_FILL_AA) != 0) { slowMask |= 1ull << c; }
				const uint32_t jp = j > 0 ? j - 1 : r.N - 1;
				const uint32_t jn = j + 1 < r.N ? j + 1 : 0;
#ifdef VGX_EXP_NOLOAD /* tuning builds only */
				v1[c] = make_float2((float)(j * 7u & 1023u), (float)(r.poly & 1023u)); vP[c] = make_float2((float)(jp * 7u & 1023u), (float)(r.poly >> 3 & 1023u)); vN[c] = make_float2((float)(jn * 5u & 1023u), (float)(r.poly >> 5 & 1023u));
#else
				v1[c] = heap[(uint64_t)r.poly + j];
				vP[c] = heap[(uint64_t)r.poly + jp];
				vN[c] = heap[(uint64_t)r.poly + jn];
#endif
			}
		}
		// (B) geometry + stores
		float* const posBase = A.pos + 2 * winV0;
		uint32_t* const colBase = A.color + winV0;
		uint16_t* const idxBase = A.idx + winI0;
#pragma unroll
		for (int c = 0; c < RUN; ++c) {
			if (c < n) {
				if ((slowMask >> c) & 1ull) { // a non-AA fill in the chunk: the generic path does the whole chunk
					(void)fill_chunk_slow(A, pos + (uint64_t)c * VGX_WAVE, elemEnd, wbase, numMeshes, lane);
					continue;
				}
				if (val[c]) {
					const Fill2Rec r = rec[c];
					const uint32_t j = jj[c], N = r.N;
					const V2 p1 = v2(v1[c].x, v1[c].y);
					const V2 d01 = v2dir_fast(v2(vP[c].x, vP[c].y), p1);
					const V2 d12 = v2dir_fast(p1, v2(vN[c].x, vN[c].y));
					const V2 vaa = v2mul(v2extrude_fast(d01, d12), r.aa);
					const V2 vin = v2add(p1, vaa), vout = v2sub(p1, vaa);
#ifdef VGX_EXP_DENSE /* tuning experiment: the three streams written densely in element order (wrong results) */
					const uint64_t eiD = pos + (uint64_t)c * VGX_WAVE + (uint64_t)lane;
					const uint32_t gv = (uint32_t)(2 * (eiD - winE0));
					float* const posBase = A.pos + 4 * winE0; uint32_t* const colBase = A.color + 2 * winE0; uint16_t* const idxBase = A.idx + 9 * winE0;
#else
					const uint32_t gv = r.firstV + 2 * j;
#endif
					PosPair pp; pp.x0 = vin.x; pp.y0 = vin.y; pp.x1 = vout.x; pp.y1 = vout.y;
					VGX_ST_GUARD(__float_as_uint(pp.x0) ^ __float_as_uint(pp.y1)) { *(PosPair*)(posBase + 2 * (size_t)gv) = pp; }
					ColPair cp; cp.c0 = r.color; cp.c1 = r.color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
					VGX_ST_GUARD(cp.c0) { *(ColPair*)(colBase + gv) = cp; }
					// indices: my nine positions [9j, 9j+9) are three whole triangles T = 3j + g: T < N-2 is fan triangle
					// (0, 2T+2, 2T+4) (stroker.cpp:769-776), else fringe triangle F = T-(N-2) = half (F&1) of the quad on edge
					// F>>1: (fb, fb+1, nextOuter) / (fb, nextOuter, nextInner) with fb = 2*edge (stroker.cpp:779-795)
					uint32_t v9[9];
#pragma unroll
					for (uint32_t g = 0; g < 3; ++g) {
						const uint32_t T = 3 * j + g;
						const bool isFan = T + 2 < N;
						const uint32_t Fq = T + 2 - N; // wraps for fan triangles, unused there
						const uint32_t ed = Fq >> 1;
						const bool second = (Fq & 1u) != 0;
						const uint32_t fb = 2 * ed;
						const bool lastEdge = ed + 1 == N;
						const uint32_t nextInner = lastEdge ? 0u : fb + 2, nextOuter = lastEdge ? 1u : fb + 3;
						v9[3 * g] = ((isFan ? 0u : fb) + r.ibase) & 0xFFFFu;
						v9[3 * g + 1] = ((isFan ? 2 * T + 2 : (second ? nextOuter : fb + 1)) + r.ibase) & 0xFFFFu;
						v9[3 * g + 2] = ((isFan ? 2 * T + 4 : (second ? nextInner : nextOuter)) + r.ibase) & 0xFFFFu;
					}
#ifdef VGX_EXP_DENSE
					uint16_t* pi = idxBase + (size_t)(9 * (eiD - winE0));
#else
					uint16_t* pi = idxBase + (size_t)(r.firstI + 9 * j);
#endif
					if (j + 1 < N) {
						Idx9 q; q.a = v9[0] | (v9[1] << 16); q.b = v9[2] | (v9[3] << 16); q.c = v9[4] | (v9[5] << 16); q.d = v9[6] | (v9[7] << 16); q.e = (uint16_t)v9[8];
						VGX_ST_GUARD(q.a ^ q.b ^ q.c ^ q.d ^ q.e) { *(Idx9*)pi = q; }
					} else {
						Idx3 q; q.a = v9[0] | (v9[1] << 16); q.b = (uint16_t)v9[2];
						*(Idx3*)pi = q;
					}
				}
			}
		}
		// owner of the run's last element: where the next window (if one is needed) starts
		{
			const uint64_t runEnd = pos + nn * VGX_WAVE < elemEnd ? pos + nn * VGX_WAVE : elemEnd;
			const int lastValid = (int)(runEnd - (pos + (nn - 1) * VGX_WAVE)) - 1;
			int kL = 0;
#pragma unroll
			for (int c = 0; c < RUN; ++c) { if (c == n - 1) { kL = wave_bcast(kk[c], lastValid); } }
			mcur = wbase + (uint64_t)kL;
		}
		pos += nn * VGX_WAVE;
	}
}

// ------------------------------------------------------------------------------------------------
// k_fill3: k_fill2's element code, software-pipelined so that a wave NEVER waits for its own stores.
// On gfx9-family parts loads and stores share one in-order counter (vmcnt): `s_waitcnt vmcnt(n)` for a load also waits for
// every store issued BEFORE that load. k_fill / k_fill2 request a chunk's vertices after the previous chunk's stores, so the
// wait for the vertices drains those stores (their write acknowledgement from L2 takes microseconds under load): per wave the
// chunk time becomes VALU + full store latency, and the kernel's "memory bound" was 3 TB/s of stores with the VALU idle
// meanwhile (measured: k_fill2 without stores 0.77 ms, without loads 2.0 ms, both 2.4 ms; the same stores issued without a
// dependent load in between run at 5.5 TB/s, profiles/micro). Here the vertices of run r+1 are requested BEFORE run r is
// emitted: when run r+1's vertices are first used, the only operations that must have completed are the stores of run r-1 --
// issued a whole run earlier -- and the compiler's wait is vmcnt(<stores of run r>), not vmcnt(0). That needs straight-line
// code between the request and the use (the compiler counts instructions, any divergent branch with a store in it makes it
// fall back to vmcnt(0)): the steady-state loop only takes FAST runs (RUN full chunks, all AA fills, inside the window,
// 32-bit heap indices); everything else goes through the generic chunk code of k_fill2.
// ------------------------------------------------------------------------------------------------
template<int RUN>
struct Fill3Run // one staged run: everything its emit needs, in registers; the vertex loads are in flight
{
	uint32_t j[RUN], N[RUN], gv[RUN], gi[RUN], color[RUN], ibase[RUN];
	float aa[RUN];
	float2 vP[RUN], v1[RUN], vN[RUN];
	float* posBase; uint32_t* colBase; uint16_t* idxBase; // wave-uniform
};

struct Fill3Win // window state: wave-uniform bases + one 32-bit relative prefix per lane
{
	uint64_t wbase, E0, V0, I0;
	uint32_t wrel, wlast;
	bool have, fast; // fast: every mesh of the window is an AA fill (or empty) with 32-bit heap indices
};

__device__ __forceinline__ void fill3_window_load(const VgxStrokeArgs& A, Fill3Win& W, Fill2Rec* s_win, uint64_t wbase, uint64_t numMeshes, int lane)
{
	W.wbase = wbase;
	const uint64_t idx = wbase + (uint64_t)lane;
	const uint64_t prefix = (idx <= numMeshes) ? A.elem_prefix[idx] : ~0ull;
	Fill2Rec r;
	r.poly = 0; r.N = 0; r.firstV = 0; r.firstI = 0; r.color = 0; r.aa = 0.0f; r.ibase = 0; r.kind = VGX_MESH_FILL_AA;
	uint64_t fv = 0, fi = 0;
	bool odd = false;
	if (idx < numMeshes) {
		const VgxMeshDesc md = A.mdesc[idx];
		const VgxMeshPrep pr = A.mprep[idx];
		r.poly = (uint32_t)md.poly_first; r.N = md.poly_n; r.kind = VGX_MD_KIND(md.kind);
		r.color = pr.color; r.aa = pr.f0;
		fv = A.mtab[idx].first_vertex;
		fi = A.mtab[idx].first_index;
		if (A.mesh_base) { r.ibase = A.mesh_base[idx]; }
		const uint64_t nextPrefix = A.elem_prefix[idx + 1];
		const bool isFillEntry = nextPrefix != prefix; // entries of the other class are zero-length in this prefix
		odd = ((md.poly_first + md.poly_n) >> 32) != 0 || (isFillEntry && r.kind != VGX_MESH_FILL_AA);
	}
	W.E0 = wave_bcast_u64(prefix, 0);
	W.V0 = wave_bcast_u64(fv, 0);
	W.I0 = wave_bcast_u64(fi, 0);
	r.firstV = (uint32_t)(fv - W.V0); r.firstI = (uint32_t)(fi - W.I0);
	const uint64_t d = prefix - W.E0;
	W.wrel = (prefix == ~0ull || d > 0x7FFFFFFFull) ? 0x7FFFFFFFu : (uint32_t)d;
	W.wlast = wave_bcast_u32(W.wrel, VGX_WAVE - 1);
	W.fast = wave_ballot(odd) == 0;
	__syncthreads(); // one-wave workgroup: lanes may still be reading the previous window
	s_win[lane] = r;
	__syncthreads();
	W.have = true;
}

// Owner search of one FULL chunk at window-relative element pc (the window covers it): window entry and element index.
__device__ __forceinline__ void fill3_owner(const Fill3Win& W, uint32_t* s_cnt, uint32_t pc, int lane, int* kOut, uint32_t* jOut)
{
	const int rel = (int)(W.wrel - pc);              // where my window entry begins relative to the chunk (<= 0: before / at its start)
	const int count0 = __popcll(wave_ballot(rel <= 0));
	if (rel > 0 && rel < VGX_WAVE) { atomicAdd(&s_cnt[rel], 1u); }
	__builtin_amdgcn_wave_barrier();
	const uint32_t heads = s_cnt[lane];
	__builtin_amdgcn_wave_barrier();
	s_cnt[lane] = 0;
	const int k = count0 - 1 + (int)wave_incl_scan_u32(heads, lane);
	const int relk = __shfl(rel, k);
	*kOut = k;
	*jOut = (uint32_t)(lane - relk);
}

template<int RUN>
__device__ __forceinline__ void fill3_stage(const VgxStrokeArgs& A, const Fill3Win& W, const Fill2Rec* s_win, uint32_t* s_cnt, uint32_t p0, int lane, Fill3Run<RUN>& R)
{
	const float2* const heap = (const float2*)A.poly;
	R.posBase = A.pos + 2 * W.V0; R.colBase = A.color + W.V0; R.idxBase = A.idx + W.I0;
#pragma unroll
	for (int c = 0; c < RUN; ++c) {
		int k; uint32_t j;
		fill3_owner(W, s_cnt + c * VGX_WAVE, p0 + (uint32_t)c * VGX_WAVE, lane, &k, &j);
		const Fill2Rec r = s_win[k];
		const uint32_t jp = j > 0 ? j - 1 : r.N - 1;
		const uint32_t jn = j + 1 < r.N ? j + 1 : 0;
		R.v1[c] = heap[(uint64_t)r.poly + j];
		R.vP[c] = heap[(uint64_t)r.poly + jp];
		R.vN[c] = heap[(uint64_t)r.poly + jn];
		R.j[c] = j; R.N[c] = r.N; R.gv[c] = r.firstV + 2 * j; R.gi[c] = r.firstI + 9 * j;
		R.color[c] = r.color; R.ibase[c] = r.ibase; R.aa[c] = r.aa;
	}
}

// One AA fill corner: two vertices, two colours, nine (last corner: three) indices. Straight-line code.
__device__ __forceinline__ void fill3_corner(float* posBase, uint32_t* colBase, uint16_t* idxBase, uint32_t j, uint32_t N, uint32_t gv, uint32_t gi,
	uint32_t color, uint32_t ibase, float aa, float2 fP, float2 f1, float2 fN)
{
	const V2 p1 = v2(f1.x, f1.y);
	const V2 d01 = v2dir_fast(v2(fP.x, fP.y), p1);
	const V2 d12 = v2dir_fast(p1, v2(fN.x, fN.y));
	const V2 vaa = v2mul(v2extrude_fast(d01, d12), aa);
	const V2 vin = v2add(p1, vaa), vout = v2sub(p1, vaa);
	PosPair pp; pp.x0 = vin.x; pp.y0 = vin.y; pp.x1 = vout.x; pp.y1 = vout.y;
	VGX_ST_GUARD(__float_as_uint(pp.x0) ^ __float_as_uint(pp.y1)) { *(PosPair*)(posBase + 2 * (size_t)gv) = pp; }
	ColPair cp; cp.c0 = color; cp.c1 = color & 0x00FFFFFFu; // colorSetAlpha(color, 0), vg.inl:95-98
	VGX_ST_GUARD(cp.c0) { *(ColPair*)(colBase + gv) = cp; }
	// indices: my nine positions [9j, 9j+9) are three whole triangles T = 3j + g: T < N-2 is fan triangle (0, 2T+2, 2T+4)
	// (stroker.cpp:769-776), else fringe triangle F = T-(N-2) = half (F&1) of the quad on edge F>>1: (fb, fb+1, nextOuter) /
	// (fb, nextOuter, nextInner) with fb = 2*edge (stroker.cpp:779-795)
	uint32_t v9[9];
#pragma unroll
	for (uint32_t g = 0; g < 3; ++g) {
		const uint32_t T = 3 * j + g;
		const bool isFan = T + 2 < N;
		const uint32_t Fq = T + 2 - N; // wraps for fan triangles, unused there
		const uint32_t ed = Fq >> 1;
		const bool second = (Fq & 1u) != 0;
		const uint32_t fb = 2 * ed;
		const bool lastEdge = ed + 1 == N;
		const uint32_t nextInner = lastEdge ? 0u : fb + 2, nextOuter = lastEdge ? 1u : fb + 3;
		v9[3 * g] = ((isFan ? 0u : fb) + ibase) & 0xFFFFu;
		v9[3 * g + 1] = ((isFan ? 2 * T + 2 : (second ? nextOuter : fb + 1)) + ibase) & 0xFFFFu;
		v9[3 * g + 2] = ((isFan ? 2 * T + 4 : (second ? nextInner : nextOuter)) + ibase) & 0xFFFFu;
	}
	uint16_t* pi = idxBase + (size_t)gi;
	if (j + 1 < N) {
		Idx9 q; q.a = v9[0] | (v9[1] << 16); q.b = v9[2] | (v9[3] << 16); q.c = v9[4] | (v9[5] << 16); q.d = v9[6] | (v9[7] << 16); q.e = (uint16_t)v9[8];
		VGX_ST_GUARD(q.a ^ q.b ^ q.c ^ q.d ^ q.e) { *(Idx9*)pi = q; }
	} else {
		Idx3 q; q.a = v9[0] | (v9[1] << 16); q.b = (uint16_t)v9[2];
		VGX_ST_GUARD(q.a ^ q.b) { *(Idx3*)pi = q; }
	}
}

template<int RUN>
__device__ __forceinline__ void fill3_emit(const Fill3Run<RUN>& R)
{
#pragma unroll
	for (int c = 0; c < RUN; ++c) {
		fill3_corner(R.posBase, R.colBase, R.idxBase, R.j[c], R.N[c], R.gv[c], R.gi[c], R.color[c], R.ibase[c], R.aa[c], R.vP[c], R.v1[c], R.vN[c]);
	}
}

// One chunk through the generic (not pipelined) code: partial chunks, non-AA fills, windows the fast path cannot take.
__device__ __forceinline__ uint64_t fill3_generic_chunk(const VgxStrokeArgs& A, const Fill3Win& W, const Fill2Rec* s_win, uint32_t* s_cnt, uint64_t pos, uint64_t elemEnd, uint64_t numMeshes, int lane)
{
	const uint32_t pc = (uint32_t)(pos - W.E0);
	int k; uint32_t j;
	fill3_owner(W, s_cnt, pc, lane, &k, &j);
	const bool valid = pos + (uint64_t)lane < elemEnd;
	const Fill2Rec r = s_win[k];
	const int nvalid = (int)((elemEnd - pos) < (uint64_t)VGX_WAVE ? (elemEnd - pos) : (uint64_t)VGX_WAVE);
	const uint64_t owner = W.wbase + (uint64_t)wave_bcast(k, nvalid - 1);
	if (!W.fast && wave_ballot(valid && (r.kind != VGX_MESH_FILL_AA)) != 0) { // a non-AA fill (or a far mesh) in the chunk
		(void)fill_chunk_slow(A, pos, elemEnd, W.wbase, numMeshes, lane);
		return owner;
	}
	if (!W.fast) { // far meshes (heap index >= 2^32): the slow path addresses them with 64 bits
		const VgxMeshDesc md = A.mdesc[W.wbase + (uint64_t)(valid ? k : 0)];
		if (wave_ballot(valid && ((md.poly_first + md.poly_n) >> 32) != 0) != 0) {
			(void)fill_chunk_slow(A, pos, elemEnd, W.wbase, numMeshes, lane);
			return owner;
		}
	}
	if (valid) {
		const float2* const heap = (const float2*)A.poly;
		const uint32_t jp = j > 0 ? j - 1 : r.N - 1;
		const uint32_t jn = j + 1 < r.N ? j + 1 : 0;
		const float2 f1 = heap[(uint64_t)r.poly + j], fP = heap[(uint64_t)r.poly + jp], fN = heap[(uint64_t)r.poly + jn];
		fill3_corner(A.pos + 2 * W.V0, A.color + W.V0, A.idx + W.I0, j, r.N, r.firstV + 2 * j, r.firstI + 9 * j, r.color, r.ibase, r.aa, fP, f1, fN);
	}
	return owner;
}

template<int RUN>
__global__ __launch_bounds__(VGX_WAVE) void k_fill3(VgxStrokeArgs A)
{
	__shared__ Fill2Rec s_win[VGX_WAVE];
	__shared__ uint32_t s_cnt[RUN * VGX_WAVE];
	const int lane = threadIdx.x;
	if (A.totals->status != VGX_OK) {
		return;
	}
	const uint64_t numMeshes = A.totals->sizes.num_meshes;
	const uint64_t totalElems = A.elem_prefix[numMeshes];
	const uint64_t numSegments = (totalElems + VGX_WAVE - 1) / VGX_WAVE;
	const uint64_t segsPerWave = (numSegments + gridDim.x - 1) / gridDim.x;
	const uint64_t seg0 = (uint64_t)blockIdx.x * segsPerWave;
	const uint64_t seg1 = (seg0 + segsPerWave < numSegments) ? seg0 + segsPerWave : numSegments;
	if (seg0 >= seg1) {
		return;
	}
	uint64_t pos = seg0 * VGX_WAVE;
	const uint64_t elemEnd = (seg1 * VGX_WAVE < totalElems) ? seg1 * VGX_WAVE : totalElems;
#pragma unroll
	for (int c = 0; c < RUN; ++c) { s_cnt[c * VGX_WAVE + lane] = 0; }
	uint64_t mcur = find_owner_u64(A.elem_prefix, 0, numMeshes, pos); // last mesh with prefix <= pos
	Fill3Win W;
	W.wbase = 0; W.E0 = 0; W.V0 = 0; W.I0 = 0; W.wrel = 0; W.wlast = 0; W.have = false; W.fast = false;
	const uint32_t runElems = RUN * VGX_WAVE;
	// a FAST run at pos: RUN full chunks inside [pos, elemEnd) and inside a fast window
#define FILL3_CAN_FAST(pos_) (W.have && W.fast && (elemEnd - (pos_)) >= runElems && W.wlast > (uint32_t)((pos_) - W.E0) + (runElems - 1))

	while (pos < elemEnd) { // wave-uniform
		if (!W.have || !(W.wlast > (uint32_t)(pos - W.E0) + (VGX_WAVE - 1))) { // the window does not cover the next chunk
			fill3_window_load(A, W, s_win, mcur, numMeshes, lane);
			if (!(W.wlast > (uint32_t)(pos - W.E0) + (VGX_WAVE - 1))) { // more than 63 mesh records inside one chunk
				mcur = fill_chunk_slow(A, pos, elemEnd, W.wbase, numMeshes, lane);
				pos += VGX_WAVE;
				W.have = false;
				continue;
			}
		}
		if (!FILL3_CAN_FAST(pos)) {
			mcur = fill3_generic_chunk(A, W, s_win, s_cnt, pos, elemEnd, numMeshes, lane);
			pos += VGX_WAVE;
			continue;
		}
		// pipelined: stage run r+1 (its vertex loads go out), then emit run r (its stores go out); two register sets
		Fill3Run<RUN> Ra, Rb;
		fill3_stage<RUN>(A, W, s_win, s_cnt, (uint32_t)(pos - W.E0), lane, Ra);
		pos += runElems;
		for (;;) {
			if (!FILL3_CAN_FAST(pos)) { fill3_emit<RUN>(Ra); break; }
			fill3_stage<RUN>(A, W, s_win, s_cnt, (uint32_t)(pos - W.E0), lane, Rb);
			pos += runElems;
			fill3_emit<RUN>(Ra);
			if (!FILL3_CAN_FAST(pos)) { fill3_emit<RUN>(Rb); break; }
			fill3_stage<RUN>(A, W, s_win, s_cnt, (uint32_t)(pos - W.E0), lane, Ra);
			pos += runElems;
			fill3_emit<RUN>(Rb);
		}
		// owner of the last element emitted: where the next window (if one is needed) starts
		{
			const uint64_t last = pos - 1;
			const uint32_t lrel = (uint32_t)(last - W.E0);
			const int cnt = __popcll(wave_ballot(W.wrel <= lrel));
			mcur = W.wbase + (uint64_t)(cnt - 1);
		}
	}
#undef FILL3_CAN_FAST
}

// ------------------------------------------------------------------------------------------------
// k_stroke: strokerPolylineStroke / StrokeAA / StrokeAAThin (stroker.cpp:1008-2314)
// ------------------------------------------------------------------------------------------------
// Window of 64 consecutive mesh records kept in LDS (64 B each): loaded cooperatively (one mesh per lane) when the
// walk leaves the previous window, read back by the element lanes with four ds_read_b128. This takes the per-element
// mesh-table gathers (a chain of dependent global loads per chunk: prefix -> descriptor -> vertex) off the critical
// path: the only global load a chunk waits for is its polyline vertices.
#ifndef VGX_STROKE_OCC
#define VGX_STROKE_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
struct __attribute__((aligned(16))) StrokeRec
{
	uint64_t polyFirst;
	uint32_t N, kind;
	uint32_t draw;
	float hsw, hswAA, fringe;
	uint64_t firstV, firstI;
	uint32_t color, ibase, pad1, pad2; // ibase: assembly index base of the mesh (0 when not armed)
};

__global__ __launch_bounds__(VGX_WAVE) VGX_STROKE_OCC void k_stroke(VgxStrokeArgs A)
{
	__shared__ StrokeRec s_win[VGX_WAVE];
	const int lane = threadIdx.x;
	if (A.totals->status != VGX_OK) {
		return;
	}
	const uint64_t numMeshes = A.totals->sizes.num_meshes;
	const uint64_t totalElems = A.elem_prefix[numMeshes];
	const uint64_t segItems = vgx_segment_items(totalElems, gridDim.x);
	const uint64_t numSegments = (totalElems + segItems - 1) / segItems;
	const uint64_t segsPerWave = (numSegments + gridDim.x - 1) / gridDim.x;
	const uint64_t seg0 = (uint64_t)blockIdx.x * segsPerWave;
	const uint64_t seg1 = (seg0 + segsPerWave < numSegments) ? seg0 + segsPerWave : numSegments;
	if (seg0 >= seg1) {
		return;
	}
	uint64_t mNext = lower_bound_u64(A.elem_prefix, 0, numMeshes, seg0 * segItems);
	uint64_t wbase = ~0ull; // first mesh of the LDS window (none yet)
	uint64_t wv = 0;        // elem_prefix[wbase + lane], ~0 past the table
	for (uint64_t seg = seg0; seg < seg1; ++seg) {
		const uint64_t m0 = mNext;
		const uint64_t m1 = advance_lower_bound(A.elem_prefix, m0, numMeshes, (seg + 1) * segItems, lane);
		mNext = m1;
		if (m0 == m1) {
			continue;
		}
		const uint64_t E0 = A.elem_prefix[m0];
		const uint64_t E1 = A.elem_prefix[m1];
		StrokeCarry carry;
		carry.v = 0; carry.i = 0; carry.rails = 0;
		uint64_t mcur = m0; // mesh that owns the chunk's first element

		for (uint64_t chunk = E0; chunk < E1; chunk += VGX_WAVE) {
			const uint64_t ei = chunk + lane;
			const bool valid = ei < E1;
			const uint64_t lastKey = chunk + (VGX_WAVE - 1);
			if (wbase == ~0ull || mcur < wbase || !(wave_bcast_u64(wv, VGX_WAVE - 1) > lastKey)) { // wave-uniform
				wbase = mcur;
				const uint64_t widx = wbase + (uint64_t)lane;
				wv = (widx <= numMeshes) ? A.elem_prefix[widx] : ~0ull;
				StrokeRec r;
				r.polyFirst = 0; r.N = 2; r.kind = VGX_MESH_STROKE_AA; r.draw = 0; r.hsw = 0.0f; r.hswAA = 0.0f; r.fringe = 1.0f;
				r.firstV = 0; r.firstI = 0; r.color = 0; r.ibase = 0; r.pad1 = 0; r.pad2 = 0;
				if (widx < numMeshes) {
					const VgxMeshDesc md = A.mdesc[widx];
					const VgxMeshPrep pr = A.mprep[widx];
					const vgx_mesh mr = A.mtab[widx];
					r.polyFirst = md.poly_first; r.N = md.poly_n; r.kind = md.kind; r.draw = md.draw;
					r.hsw = pr.f0; r.hswAA = pr.f1; r.fringe = pr.f2; r.color = pr.color;
					r.firstV = mr.first_vertex; r.firstI = mr.first_index;
					if (A.mesh_base) { r.ibase = A.mesh_base[widx]; }
				}
				__syncthreads(); // lanes may still be reading the previous window
				s_win[lane] = r;
				__syncthreads();
			}
			const bool windowCovers = wave_bcast_u64(wv, VGX_WAVE - 1) > lastKey;
			const uint32_t wrel = window_rel(wv, chunk);
			const int ownerOfs = window_owner_rel(wrel, valid ? (uint32_t)lane : 0u);
			const uint32_t orel = (uint32_t)__shfl((int)wrel, ownerOfs);
			const int firstOwner = __popcll(wave_ballot(wv <= chunk)) - 1;
			uint64_t ownerBase = orel > 0 ? chunk + orel : wave_bcast_u64(wv, firstOwner < 0 ? 0 : firstOwner);
			uint64_t mi = m0;
			MeshCtx mc;
			mc.kind = VGX_MESH_STROKE_AA; mc.N = 2; mc.j = 0; mc.cap = 0; mc.join = 0; mc.closed = false;
			mc.hsw = 0.0f; mc.hswAA = 0.0f; mc.fringe = 1.0f; mc.dr = A.draws; mc.vtx.p = (const float2*)A.poly;
			uint32_t color = 0;
			uint64_t firstV = 0, firstI = 0;
			uint32_t idxBase = 0;
			if (valid) {
				StrokeRec r;
				if (windowCovers) {
					mi = wbase + (uint64_t)ownerOfs;
					r = s_win[ownerOfs];
				} else { // the window is full of zero-length (fill) entries: rare, fall back to a search
					mi = find_owner_u64(A.elem_prefix, m0, m1, ei);
					ownerBase = A.elem_prefix[mi];
					const VgxMeshDesc md = A.mdesc[mi];
					const VgxMeshPrep pr = A.mprep[mi];
					r.polyFirst = md.poly_first; r.N = md.poly_n; r.kind = md.kind; r.draw = md.draw;
					r.hsw = pr.f0; r.hswAA = pr.f1; r.fringe = pr.f2; r.color = pr.color;
					r.firstV = A.mtab[mi].first_vertex; r.firstI = A.mtab[mi].first_index;
					r.ibase = A.mesh_base ? A.mesh_base[mi] : 0u;
				}
				mc.kind = VGX_MD_KIND(r.kind);
				mc.closed = VGX_MD_CLOSED(r.kind) != 0;
				mc.cap = VGX_MD_CAP(r.kind);
				mc.join = VGX_MD_JOIN(r.kind);
				mc.N = r.N;
				mc.j = (uint32_t)(ei - ownerBase);
#ifdef VGX_EXP_WRAPREAD
				mc.vtx.p = (const float2*)A.poly + (r.polyFirst & 0x3FFFFull);
#else
				mc.vtx.p = (const float2*)A.poly + r.polyFirst;
#endif
				mc.hsw = r.hsw; mc.hswAA = r.hswAA; mc.fringe = r.fringe;
				mc.dr = A.draws + r.draw;
				color = r.color;
				firstV = r.firstV; firstI = r.firstI; idxBase = r.ibase;
			}
			const int nvalid = (int)((E1 - chunk) < (uint64_t)VGX_WAVE ? (E1 - chunk) : (uint64_t)VGX_WAVE);
			const int Lz = nvalid - 1;
			stroke_chunk(valid, lane < VGX_WAVE - 1 && ei + 1 < E1, nvalid, lane, mc, color, A.pos + 2 * firstV, A.color + firstV, A.idx + firstI, idxBase, carry);
			mcur = wave_bcast_u64(mi, Lz);
		}
	}
}

// The caller's mesh table = the internal one once the scan over meshes has filled first_vertex / first_index.
__global__ __launch_bounds__(256) void k_copy_meshes(VgxStrokeArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	const uint64_t n = A.totals->sizes.num_meshes * (sizeof(vgx_mesh) / sizeof(uint4));
	const uint4* src = (const uint4*)A.mtab;
	uint4* dst = (uint4*)A.meshes_out;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		dst[i] = src[i];
	}
}

} // namespace

void vgx_launch_mesh_prepare(const VgxStrokeArgs& a, hipStream_t s)
{
	hipLaunchKernelGGL(k_mesh_prepare, dim3(2048), dim3(256), 0, s, a);
}

void vgx_launch_fill(const VgxStrokeArgs& a, int numBlocks, hipStream_t s, int variant)
{
	if (a.meshes_out) { // the caller's mesh table, in one streaming copy (not a dependent load + store inside every chunk)
		hipLaunchKernelGGL(k_copy_meshes, dim3(1024), dim3(256), 0, s, a);
	}
	switch (variant) { // 0 = k_fill (round 1 / 2), n = k_fill2 with runs of n chunks
	case 1: hipLaunchKernelGGL(k_fill2<1>, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a); break;
	case 2: hipLaunchKernelGGL(k_fill2<2>, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a); break;
	case 4: hipLaunchKernelGGL(k_fill2<4>, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a); break;
	case 31: hipLaunchKernelGGL(k_fill3<1>, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a); break;
	case 32: hipLaunchKernelGGL(k_fill3<2>, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a); break;
	case 34: hipLaunchKernelGGL(k_fill3<4>, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a); break;
	default: hipLaunchKernelGGL(k_fill, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a); break;
	}
}

void vgx_launch_stroke(bool emit, const VgxStrokeArgs& a, int numBlocks, hipStream_t s)
{
	if (emit) {
		hipLaunchKernelGGL(k_stroke, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
	} else { // sizes Round-join meshes; returns at once when the batch has none (every other size is closed-form)
		hipLaunchKernelGGL(k_round_sizes, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
	}
}
