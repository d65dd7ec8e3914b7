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
#include "vgx_tile.h"

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
	uint32_t pad;      // 1: VGX_FILL_INDEX_ORDER_SSE
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
		r.polyFirst = md.poly_first; r.N = md.poly_n; r.kind = VGX_MD_KIND(md.kind); r.pad = VGX_MD_SSE_ORDER(md.kind);
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
	F.sseOrder = r.pad != 0;
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
	F.aaElem = false; F.prevInWave = false; F.nextInWave = false; F.sseOrder = false;
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
		F.sseOrder = VGX_MD_SSE_ORDER(md.kind) != 0;
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

// The fill elements [pos, elemEnd) of the flat fill-element stream (A.elem_prefix = the fills' prefix), mcur = the mesh that
// owns `pos` (last mesh with prefix <= pos). W / wbase persist between calls of one wave (k_emit calls this once per run of
// fill meshes): the window stays valid while it covers the next chunk. Whole meshes are NOT required (a fill element only
// needs its own mesh record and its two neighbours), the walk is a plain 64-element stride.
__device__ __forceinline__ void fill_range(const VgxStrokeArgs& A, FillWindow& W, uint64_t& wbase, float2* s_ring, uint64_t pos, const uint64_t elemEnd, uint64_t mcur,
	const uint64_t numMeshes, const int lane)
{
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

#ifndef VGX_FILL_OCC
#define VGX_FILL_OCC
#endif
__global__ __launch_bounds__(VGX_WAVE) VGX_FILL_OCC void k_fill(VgxStrokeArgs A)
{
	__shared__ FillRec s_win[VGX_WAVE];
	__shared__ uint64_t s_pre[VGX_WAVE];
	__shared__ float2 s_ring[VGX_FILL_RING + 4];
	const int lane = threadIdx.x;
	if (A.totals->status != VGX_OK || (A.tile_mode && vgx_tile_mode_on(A.totals))) { // (k_emit_tiles, vgx_tile.hip, writes such a batch's fills and strokes)
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
	// This wave's elements are the contiguous range [pos, elemEnd)
	const uint64_t pos = seg0 * VGX_WAVE;
	const uint64_t elemEnd = (seg1 * VGX_WAVE < totalElems) ? seg1 * VGX_WAVE : totalElems;
	const uint64_t mcur = find_owner_u64(A.elem_prefix, 0, numMeshes, pos); // last mesh with prefix <= pos
	uint64_t wbase = mcur;
	FillWindow W;
	W.rec = s_win; W.pre = s_pre;
	W.prefix = 0; // no window yet: the first chunk loads one
	fill_range(A, W, wbase, s_ring, pos, elemEnd, mcur, numMeshes, lane);
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

// The stroke meshes [m0, m1) (whole meshes; A.elem_prefix = the strokes' prefix): elements walked in 64-element chunks with
// the carries of a mesh that spans chunks. wbase / wv = the LDS window of 64 mesh records (s_win), kept between calls.
#define VGX_STROKE_STAGE_COL 704   /* k_stroke_long: colours (vertices) and indices of one chunk the LDS stage holds; a chunk that needs more is stored per lane */
#define VGX_STROKE_STAGE_IDX 3072
template<bool ONLY_SIMPLE, int SCOL, int SIDX>
__device__ __forceinline__ void stroke_range(const VgxStrokeArgs& A, StrokeRec* s_win, uint64_t& wbase, uint64_t& wv, const uint64_t m0, const uint64_t m1,
	const uint64_t numMeshes, const int lane, StrokeStageT<SCOL, SIDX>* stage)
{
	const uint64_t E0 = A.elem_prefix[m0];
	const uint64_t E1 = A.elem_prefix[m1];
	StrokeCarry carry;
	carry.v = 0; carry.i = 0; carry.rails = 0;
	uint64_t mcur = m0; // mesh that owns the chunk's first element
#ifdef VGX_STROKE_PROFILE
	carry.tw = 0; carry.tg = 0; carry.te = 0;
	const unsigned long long tr0 = clock64();
	unsigned long long nch = 0;
#endif

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
				// Round joins / caps: the mesh's arc step (stroker.cpp:1013, 1398) once per mesh here, not once per element in elem_geometry
				if (!ONLY_SIMPLE && VGX_MD_KIND(md.kind) != VGX_MESH_STROKE_AA_THIN && (VGX_MD_JOIN(md.kind) == VGX_JOIN_ROUND || (!VGX_MD_CLOSED(md.kind) && VGX_MD_CAP(md.kind) == VGX_CAP_ROUND))) {
					const vgx_draw* dr = A.draws + md.draw;
					r.pad1 = __float_as_uint(vgx_step_angle(dr->scale, pr.f0, dr->tess_tol));
					r.pad2 = 1u;
				}
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
				r.pad1 = 0; r.pad2 = 0;
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
			if (r.pad2) { mc.da = __uint_as_float(r.pad1); } // (else < 0: evaluated where it is needed)
			color = r.color;
			firstV = r.firstV; firstI = r.firstI; idxBase = r.ibase;
		}
		const int nvalid = (int)((E1 - chunk) < (uint64_t)VGX_WAVE ? (E1 - chunk) : (uint64_t)VGX_WAVE);
		const int Lz = nvalid - 1;
		if (ONLY_SIMPLE) { // k_stroke_simple: the scan over the meshes found closed Miter AA / Thin strokes only
			stroke_chunk_simple(valid, lane < VGX_WAVE - 1 && ei + 1 < E1, nvalid, lane, mc, color, A.pos + 2 * firstV, A.color + firstV, A.idx + firstI, idxBase, carry);
		} else if (wave_ballot(valid && !stroke_elem_is_simple(mc.kind, mc.closed, mc.join)) == 0) { // wave-uniform
			stroke_chunk_simple(valid, lane < VGX_WAVE - 1 && ei + 1 < E1, nvalid, lane, mc, color, A.pos + 2 * firstV, A.color + firstV, A.idx + firstI, idxBase, carry);
		} else {
			// one mesh in the whole chunk (lane 0's is everybody's): its output may go through the LDS stage (long polylines: all but the chunks at a mesh's ends)
			const bool oneMesh = stage != nullptr && wave_ballot(valid && mi != wave_bcast_u64(mi, 0)) == 0;
			stroke_chunk(valid, lane < VGX_WAVE - 1 && ei + 1 < E1, nvalid, lane, mc, color, A.pos + 2 * firstV, A.color + firstV, A.idx + firstI, idxBase, carry, stage, oneMesh);
		}
		mcur = wave_bcast_u64(mi, Lz);
#ifdef VGX_STROKE_PROFILE
		++nch;
#endif
	}
#ifdef VGX_STROKE_PROFILE
	if (lane == 0) {
		atomicAdd(&A.totals->prof[0], carry.tw); atomicAdd(&A.totals->prof[1], carry.tg); atomicAdd(&A.totals->prof[2], carry.te);
		atomicAdd(&A.totals->prof[3], clock64() - tr0); atomicAdd(&A.totals->prof[4], nch);
	}
#endif
}

// Two instantiations, both launched, one exits at once (the scan over the meshes decided: totals->has_general_stroke):
//   k_stroke         every kind of stroke (128 VGPRs, 4 waves per SIMD)
//   k_stroke_simple  batches whose strokes are all closed, Miter, AA or Thin -- e.g. the tiger: stroke_chunk_simple only, 56 VGPRs,
//                    8 waves per SIMD: 1.21 ms against 1.29 ms on the same box
// k_stroke_long (SIDX != 0) takes the batches whose stroke meshes ALL have VGX_LONG_STROKE elements or more (found by the scan over
// the meshes: long polylines, where nearly every chunk lies inside one mesh), k_stroke the others (a drawing's short sub-paths, several
// meshes per chunk: the stage would never be used and costs a wave per SIMD); frame-sized calls (A.no_long) launch k_stroke only.
template<bool ONLY_SIMPLE, int SCOL, int SIDX>
__device__ __forceinline__ void stroke_kernel_body(const VgxStrokeArgs& A, StrokeRec* s_win, StrokeStageT<SCOL, SIDX>* stage)
{
	const int lane = threadIdx.x;
	if (A.totals->status != VGX_OK || (A.totals->has_general_stroke != 0u) == ONLY_SIMPLE) {
		return;
	}
	if (!ONLY_SIMPLE && !A.no_long && (A.totals->has_short_stroke == 0u) != (SIDX != 0)) {
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
		stroke_range<ONLY_SIMPLE, SCOL, SIDX>(A, s_win, wbase, wv, m0, m1, numMeshes, lane, stage);
	}
}

__global__ __launch_bounds__(VGX_WAVE) VGX_STROKE_OCC void k_stroke(VgxStrokeArgs A)
{
	__shared__ StrokeRec s_win[VGX_WAVE];
	stroke_kernel_body<false, 0, 0>(A, s_win, (StrokeStageT<0, 0>*)nullptr);
}

// The same kernel with the LDS stage (vgx_elem.h, stroke_chunk): batches of LONG polylines, where nearly every chunk lies inside one
// mesh. 12.9 KB of LDS = three waves per SIMD, and the registers of three (no spills: at four the stage costs six).
#ifndef VGX_STROKE_LONG_OCC
#define VGX_STROKE_LONG_OCC __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
__global__ __launch_bounds__(VGX_WAVE) VGX_STROKE_LONG_OCC void k_stroke_long(VgxStrokeArgs A)
{
	__shared__ StrokeRec s_win[VGX_WAVE];
	__shared__ StrokeStageT<VGX_STROKE_STAGE_COL, VGX_STROKE_STAGE_IDX> s_stage;
	stroke_kernel_body<false, VGX_STROKE_STAGE_COL, VGX_STROKE_STAGE_IDX>(A, s_win, &s_stage);
}

__global__ __launch_bounds__(VGX_WAVE) void k_stroke_simple(VgxStrokeArgs A)
{
	__shared__ StrokeRec s_win[VGX_WAVE];
	stroke_kernel_body<true, 0, 0>(A, s_win, (StrokeStageT<0, 0>*)nullptr);
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

void vgx_launch_fill(const VgxStrokeArgs& a, int numBlocks, hipStream_t s)
{
	if (a.meshes_out) { // the caller's mesh table, in one streaming copy (not a dependent load + store inside every chunk)
		hipLaunchKernelGGL(k_copy_meshes, dim3(1024), dim3(256), 0, s, a);
	}
	hipLaunchKernelGGL(k_fill, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
}

void vgx_launch_stroke(bool emit, const VgxStrokeArgs& a, int numBlocks, hipStream_t s)
{
	if (emit) {
		if (!a.tile_mode) { hipLaunchKernelGGL(k_stroke_simple, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a); } // one of the two exits at once (tile mode: k_emit_tiles took the simple batches)
		hipLaunchKernelGGL(k_stroke, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
		if (!a.no_long) { hipLaunchKernelGGL(k_stroke_long, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a); } // (one of k_stroke / k_stroke_long exits at once)
	} else { // sizes Round-join meshes; returns at once when the batch has none (every other size is closed-form)
		hipLaunchKernelGGL(k_round_sizes, dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
	}
}
