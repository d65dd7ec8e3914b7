// vgx_flatten.hip -- batch path flattening on gfx950 (replaces vg::pathXXX, reference src/path.cpp).
//
// Work decomposition
//   The batch is a flat stream of "command instances" (draw d, command k of its path). A wavefront owns
//   a SEGMENT = all draws whose first command instance falls into one 64-instance bucket, i.e. whole
//   draws, and walks their command instances 64 at a time: one lane = one path command.
//     - MOVE_TO / LINE_TO / CUBIC_TO / QUAD_TO / POLYLINE: the lane's start point is the previous
//       command's end point, which sits right in front of the lane's own arguments (args[-2..-1]); no
//       sequential dependency. Cubics run the reference's adaptive subdivision as a per-lane DFS with
//       the 10-entry pending stack in LDS (lane-interleaved float2 -> conflict free).
//     - closed shapes (RECT, ROUNDED_RECT*, CIRCLE, ELLIPSE) are self-contained sub-paths: the lane
//       simulates the reference's builder exactly (PathSim) for its own command.
//     - the growing polyline's bookkeeping (vertex offsets inside the draw, sub-path table, pathClose,
//       per-draw totals, mesh ranks) is done with wave ballots + prefix scans segmented by draw and by
//       sub-path, with carries across 64-command chunks.
//   Two passes (template<EMIT>): count -> device-wide scan over draws -> emit. The count pass stores one
//   word per command instance (cmd_cnt) so the emit pass runs every DFS exactly once.
//
// Exactness
//   pathAddVertex's epsilon de-duplication (path.cpp:767-777) and the silent piece drop when the
//   pending stack is full (path.cpp:168-179) make "the last vertex" differ from the previous
//   command's nominal end point. Both are detected lane-locally; a draw where either happens (and any
//   path containing ARC / ARC_TO, whose end points are computed) is re-done by ONE lane running the
//   exact sequential algorithm (PathSim over the whole draw). Degenerate input is slow, never wrong.
#include <stdlib.h>
#include "vgx_internal.h"
#include "vgx_wave.h"
#include "vgx_pathsim.h"
#include "vgx_walk.h"
#include "vgx_elem.h"
#include "vgx_scan_ops.h"
#include "vgx_inst.h"
#include "vgx_flat1.h"
#include "vgx_thin.h"

namespace {

// ------------------------------------------------------------------------------------------------
// LDS levels of the pending stack in the ordered two-pass kernels (count / emit of vgx_flatten_*): they run every
// subdivision twice, have no leaf slots to pay for, and long curves (BASELINE config 1: ~46 segments per cubic, 6-7
// pending halves) would leave a 4-level stack on almost every cubic.
#ifndef VGX_FLAT_LDS_LEVELS
#define VGX_FLAT_LDS_LEVELS 6 /* measured on 1 M random cubics: 4 levels 1.93 ms per count + emit, 6: 1.45, 8: 1.51, 10: 1.66 */
#endif
template<bool EMIT, bool XFORM>
__global__ __launch_bounds__(VGX_WAVE) void k_flatten(VgxFlattenArgs A)
{
	__shared__ float2 s_stack[VGX_FLAT_LDS_LEVELS * 3 * VGX_WAVE];
	const int lane = threadIdx.x;
	LdsStackT<VGX_FLAT_LDS_LEVELS> stack;
	stack.base = &s_stack[lane];

	const VgxPathSetDev& ps = A.ps;
	const uint64_t totalCmds = A.cmd_prefix[A.ndraws];
	if (A.totals->status != VGX_OK) { return; } // capacity / range errors detected by the scan steps
	const uint64_t segItems = vgx_segment_items(totalCmds, gridDim.x);
	const uint64_t numSegments = (totalCmds + segItems - 1) / segItems;
	// contiguous run of segments per wave: one binary search per wave, then cooperative advance (vgx_wave.h)
	const uint64_t segsPerWave = (numSegments + gridDim.x - 1) / gridDim.x;
	const uint64_t seg0 = (uint64_t)blockIdx.x * segsPerWave;
	const uint64_t seg1 = (seg0 + segsPerWave < numSegments) ? seg0 + segsPerWave : numSegments;
	if (seg0 >= seg1) {
		return;
	}
	uint64_t dNext = lower_bound_u64(A.cmd_prefix, 0, A.ndraws, seg0 * segItems);
	uint64_t wbase = dNext;
	DrawWindow W = draw_window_load(A, wbase, lane);

	for (uint64_t seg = seg0; seg < seg1; ++seg) {
		const uint64_t d0 = dNext;
		const uint64_t d1 = advance_lower_bound(A.cmd_prefix, d0, A.ndraws, (seg + 1) * segItems, lane);
		dNext = d1;
		if (d0 == d1) {
			continue;
		}
		const uint64_t C0 = A.cmd_prefix[d0];
		const uint64_t C1 = A.cmd_prefix[d1];
		uint64_t dcur = d0; // draw that owns the chunk's first command instance

		// carries of the draw / sub-path that continue across 64-command chunks (wave-uniform)
		int carryDrawVerts = 0, carrySpVerts = 0, carrySubs = 0, carryFill = 0, carryStroke = 0;
		int carrySlow = 0, carrySpExists = 0;

		for (uint64_t chunk = C0; chunk < C1; chunk += VGX_WAVE) {
			const uint64_t ci = chunk + lane;
			const bool valid = ci < C1;

			// ---- decode my command instance ------------------------------------------------------
			uint64_t d = d0;
			uint32_t c = 0, type = VGX_CMD_CLOSE, cflags = 0, na = 0;
			bool drawHead = false, drawLast = false, serialDraw = false;
			float scale = 1.0f, tol = 0.25f;
			uint32_t fillFlags = 0, strokeFlags = 0;
			const vgx_draw* dr = A.draws;
			// owner draw of every lane from the lane-resident draw window (no global binary search)
			const uint64_t lastKey = chunk + (VGX_WAVE - 1);
			if (!(wave_bcast_u64(W.prefix, VGX_WAVE - 1) > lastKey)) {
				wbase = dcur;
				W = draw_window_load(A, wbase, lane);
			}
			const bool windowCovers = wave_bcast_u64(W.prefix, VGX_WAVE - 1) > lastKey;
			// owner = last window entry <= my key, searched on 32-bit offsets relative to the chunk start
			const uint32_t wrel = window_rel(W.prefix, chunk);
			const int ownerOfs = window_owner_rel(wrel, valid ? (uint32_t)lane : 0u);
			const uint32_t orel = (uint32_t)__shfl((int)wrel, ownerOfs);
			const int firstOwner = __popcll(wave_ballot(W.prefix <= chunk)) - 1; // draw that owns the chunk's first command
			uint64_t ownerBase = orel > 0 ? chunk + orel : wave_bcast_u64(W.prefix, firstOwner < 0 ? 0 : firstOwner);
			uint32_t pc0 = (uint32_t)__shfl((int)(W.pc0 | ((W.serial & 1u) << 31)), ownerOfs);
			uint32_t serialStatic = pc0 >> 31;
			pc0 &= 0x7FFFFFFFu;
			VgxCmdRec rec;
			rec.type = VGX_CMD_CLOSE; rec.flags = 0; rec.na = 0; rec.arg_off = 0;
			rec.start[0] = 0.0f; rec.start[1] = 0.0f;
			for (int i = 0; i < 8; ++i) { rec.a[i] = 0.0f; }
			if (valid) {
				if (windowCovers) {
					d = wbase + (uint64_t)ownerOfs;
				} else { // more than 63 draws begin inside this chunk (1-command or empty paths)
					d = find_owner_u64(A.cmd_prefix, d0, d1, ci);
					ownerBase = A.cmd_prefix[d];
					const uint32_t path = A.draws[d].path;
					pc0 = ps.path_cmd_begin[path];
					serialStatic = ps.path_flags[path] & VGX_PF_SERIAL;
				}
				dr = A.draws + d;
				const uint32_t k = (uint32_t)(ci - ownerBase);
				c = pc0 + k;
				rec = ps.cmdrec[c]; // one 64-byte record: type, flags, start point, arguments, sub-path first point
				type = rec.type;
				cflags = rec.flags;
				na = rec.na;
				drawHead = (k == 0);
				drawLast = (cflags & VGX_CF_LAST_IN_PATH) != 0;
				scale = dr->scale;
				tol = dr->tess_tol;
				fillFlags = dr->fill_flags;
				strokeFlags = dr->stroke_flags;
				serialDraw = EMIT ? ((A.dinfo[d].flags & 1u) != 0) : (serialStatic != 0);
			}
			const float* a = rec.a;                 // arguments 0..7 (CLOSE: a[6..7] = its sub-path's first point)
			const float* pa = ps.args + rec.arg_off; // POLYLINE's variable-length arguments
			const float* mtx = dr->mtx;

			// ---- per-lane vertex count (count pass: compute; emit pass: read back) -----------------
			int cnt = 0;
			bool slow = false, exists = false, closedHere = false, pop = false;
			if (valid && !serialDraw) {
				if (!EMIT) {
					const V2 start = v2(rec.start[0], rec.start[1]); // previous command's end point (unused by sub-path starters)
					switch (type) {
					case VGX_CMD_MOVE_TO: cnt = 1; exists = true; break;
					case VGX_CMD_LINE_TO: cnt = 1; slow = v2near(start, v2(a[0], a[1])); break;
					case VGX_CMD_CUBIC_TO:
					case VGX_CMD_QUAD_TO: {
						FastCubicSink<false, false> sink;
						sink.prev = start; sink.n = 0; sink.slow = false;
						float c1x = a[0], c1y = a[1], c2x = a[2], c2y = a[3], ex, ey;
						if (type == VGX_CMD_QUAD_TO) {
							ex = a[2]; ey = a[3];
							vgx_quad_to_cubic(start.x, start.y, a[0], a[1], ex, ey, &c1x, &c1y, &c2x, &c2y);
						} else {
							ex = a[4]; ey = a[5];
						}
						wave_flatten_cubic(start.x, start.y, c1x, c1y, c2x, c2y, ex, ey, tol / (scale * scale), stack, sink);
						sink.flush();
						cnt = (int)sink.n;
						slow = sink.slow;
					} break;
					case VGX_CMD_POLYLINE: {
						const uint32_t npts = na >> 1;
						cnt = (int)npts - ((npts > 0 && v2near(start, v2(pa[0], pa[1]))) ? 1 : 0);
						slow = cnt == 0; // a fully de-duplicated polyline leaves the last vertex unchanged: not its nominal end point
					} break;
					case VGX_CMD_CLOSE: break; // decided below, needs the sub-path's vertex count
					default: break; // shapes / arcs only occur in serial paths (k_flatten_serial)
					}
				} else {
					const uint32_t w = A.cmd_cnt[ci];
					exists = (w & VGX_CC_EXISTS) != 0;
					closedHere = (w & VGX_CC_CLOSED) != 0;
					pop = (w & VGX_CC_POP) != 0;
					cnt = pop ? -1 : (int)(w & VGX_CC_COUNT_MASK);
				}
			}

			// ---- segmented bookkeeping ----------------------------------------------------------
			const uint64_t drawHeads = wave_ballot(valid && drawHead);
			const uint64_t subHeads = wave_ballot(valid && (cflags & VGX_CF_STARTS_SUB));
			const int dh = seg_head(drawHeads, lane);
			const int sh = seg_head(subHeads, lane);

			if (!EMIT) {
				// pathClose (path.cpp:707-726) needs the vertex count of its sub-path so far.
				const int incl1 = wave_incl_scan(cnt, lane);
				const int spBefore = seg_rel(incl1 - cnt, sh, carrySpVerts);
				if (valid && !serialDraw && type == VGX_CMD_CLOSE && spBefore > 2) {
					closedHere = true;
					if (v2near(v2(rec.start[0], rec.start[1]), v2(rec.a[6], rec.a[7]))) {
						pop = true;
						cnt = -1;
					}
				}
			}
			const int incl = wave_incl_scan(cnt, lane);
			const int excl = incl - cnt;
			const int inDrawBefore = seg_rel(excl, dh, carryDrawVerts);
			const int spBefore = seg_rel(excl, sh, carrySpVerts);
			const int spTotal = spBefore + cnt; // vertices of my sub-path up to and including me

			const uint64_t existMask = wave_ballot(valid && exists);
			const uint64_t mine = seg_mask_upto(dh, lane);
			const int subsIncl = __popcll(existMask & mine) + (dh < 0 ? carrySubs : 0);
			const int headExists = (sh < 0) ? carrySpExists : (int)((existMask >> sh) & 1ull);

			const bool lastInSub = valid && (cflags & VGX_CF_LAST_IN_SUB) && headExists;
			const uint64_t fillMask = wave_ballot(lastInSub && (fillFlags & VGX_FILL_ENABLE) && spTotal >= 3);
			const uint64_t strokeMask = wave_ballot(lastInSub && (strokeFlags & VGX_STROKE_ENABLE) && spTotal >= 2);
			const int fillIncl = __popcll(fillMask & mine) + (dh < 0 ? carryFill : 0);
			const int strokeIncl = __popcll(strokeMask & mine) + (dh < 0 ? carryStroke : 0);
			const uint64_t slowMask = wave_ballot(valid && slow);
			const bool slowDraw = ((slowMask & mine) != 0) || (dh < 0 && carrySlow);

			if (!EMIT) {
				if (lastInSub && spTotal > VGX_LONG_SUBPATH) { // sizing input of the single-pass heap (vgx_tessellate_count)
					atomicAdd(&A.totals->long_subpath_vertices, (unsigned long long)spTotal);
				}
				if (lastInSub && spTotal > (int)VGX_INST_LONG_SUBPATH) { // ... and of the instanced kernel's lane-private blocks
					atomicAdd(&A.totals->inst_long_subpath_vertices, (unsigned long long)spTotal);
				}
				if (valid) {
					uint32_t w = (uint32_t)(cnt < 0 ? 0 : cnt) & VGX_CC_COUNT_MASK;
					if (exists) { w |= VGX_CC_EXISTS; }
					if (closedHere) { w |= VGX_CC_CLOSED; }
					if (pop) { w |= VGX_CC_POP; }
					A.cmd_cnt[ci] = w;
				}
				if (valid && drawLast && !(serialStatic != 0)) {
					vgx_draw_info di;
					di.first_poly_vertex = 0; di.first_subpath = 0; di.first_mesh = 0;
					if (slowDraw) {
						// degenerate input (epsilon de-dup hit / dropped piece): k_flatten_serial redoes this draw exactly
						di.num_poly_vertices = 0; di.num_subpaths = 0; di.num_meshes = 0;
						di.flags = 1u;
					} else {
						di.num_poly_vertices = (uint32_t)(inDrawBefore + cnt);
						di.num_subpaths = (uint32_t)subsIncl;
						di.num_meshes = (uint32_t)(fillIncl + strokeIncl);
						di.flags = ((uint32_t)fillIncl << 1);
					}
					A.dinfo[d] = di;
				}
			} else if (valid) {
				const vgx_draw_info di = A.dinfo[d];
				if (!serialDraw) {
					// ---- vertices ---------------------------------------------------------------
					const uint64_t vbase = di.first_poly_vertex + (uint64_t)inDrawBefore;
					float* out = A.poly + 2 * vbase;
					uint32_t limit = (uint32_t)(cnt < 0 ? 0 : cnt);
					if ((cflags & VGX_CF_NEXT_IS_CLOSE) && limit > 0 && (A.cmd_cnt[ci + 1] & VGX_CC_POP)) {
						--limit; // my last vertex is the one pathClose removes
					}
					const V2 start = v2(rec.start[0], rec.start[1]);
					switch (type) {
					case VGX_CMD_MOVE_TO:
					case VGX_CMD_LINE_TO:
						if (limit > 0) {
							V2 p = v2(a[0], a[1]);
							if (XFORM) { p = v2xform(p, mtx); }
							*(float2*)out = make_float2(p.x, p.y);
						}
						break;
					case VGX_CMD_CUBIC_TO:
					case VGX_CMD_QUAD_TO: {
						FastCubicSink<true, XFORM> sink;
						sink.prev = start; sink.n = 0; sink.slow = false; sink.out = out; sink.writeLimit = limit; sink.mtx = mtx; sink.begin();
						float c1x = a[0], c1y = a[1], c2x = a[2], c2y = a[3], ex, ey;
						if (type == VGX_CMD_QUAD_TO) {
							ex = a[2]; ey = a[3];
							vgx_quad_to_cubic(start.x, start.y, a[0], a[1], ex, ey, &c1x, &c1y, &c2x, &c2y);
						} else {
							ex = a[4]; ey = a[5];
						}
						wave_flatten_cubic(start.x, start.y, c1x, c1y, c2x, c2y, ex, ey, tol / (scale * scale), stack, sink);
						sink.flush();
					} break;
					case VGX_CMD_POLYLINE: {
						const uint32_t npts = na >> 1;
						const uint32_t skip = npts - (uint32_t)(cnt < 0 ? 0 : cnt);
						for (uint32_t i = 0; i < limit; ++i) {
							V2 p = v2(pa[2 * (i + skip)], pa[2 * (i + skip) + 1]);
							if (XFORM) { p = v2xform(p, mtx); }
							*(float2*)(out + 2 * i) = make_float2(p.x, p.y);
						}
					} break;
					case VGX_CMD_CLOSE: break;
					default: break;
					}
					// ---- sub-path record + mesh descriptors, written by the sub-path's last command --
					if (lastInSub) {
						const uint32_t subIndex = (uint32_t)(subsIncl - 1);
						const uint32_t n = (uint32_t)spTotal;
						const uint64_t firstV = di.first_poly_vertex + (uint64_t)(inDrawBefore - spBefore);
						vgx_subpath r;
						r.first_vertex = firstV;
						r.num_vertices = n;
						r.flags = closedHere ? 1u : 0u;
						A.subs[di.first_subpath + subIndex] = r;
						if (A.mdesc) {
							const uint32_t numFill = di.flags >> 1;
							if ((fillFlags & VGX_FILL_ENABLE) && n >= 3) {
								vgx_write_mesh(A.mdesc, A.mtab, di.first_mesh + (uint32_t)(fillIncl - 1), dr, (uint32_t)d, subIndex, (fillFlags & VGX_FILL_AA) ? VGX_MESH_FILL_AA : VGX_MESH_FILL, closedHere, firstV, n);
							}
							if ((strokeFlags & VGX_STROKE_ENABLE) && n >= 2) {
								const uint32_t kd = !(strokeFlags & VGX_STROKE_AA) ? VGX_MESH_STROKE : ((strokeFlags & VGX_STROKE_THIN) ? VGX_MESH_STROKE_AA_THIN : VGX_MESH_STROKE_AA);
								if (vgx_write_mesh(A.mdesc, A.mtab, di.first_mesh + numFill + (uint32_t)(strokeIncl - 1), dr, (uint32_t)d, subIndex, kd, closedHere, firstV, n)) {
									atomicAdd(&A.totals->num_round_meshes, 1u);
								}
							}
						}
					}
				}
			}

			// ---- carries into the next chunk (taken from the last valid lane) ---------------------
			const int nvalid = (int)((C1 - chunk) < (uint64_t)VGX_WAVE ? (C1 - chunk) : (uint64_t)VGX_WAVE);
			const int L = nvalid - 1;
			const int lastIsDrawLast = wave_bcast((int)drawLast, L);
			const int lastIsSubLast = wave_bcast((int)((cflags & VGX_CF_LAST_IN_SUB) != 0), L);
			const int nDraw = wave_bcast(inDrawBefore + cnt, L);
			const int nSp = wave_bcast(spTotal, L);
			const int nSubs = wave_bcast(subsIncl, L);
			const int nFill = wave_bcast(fillIncl, L);
			const int nStroke = wave_bcast(strokeIncl, L);
			const int nSlow = wave_bcast((int)slowDraw, L);
			const int nHeadExists = wave_bcast(headExists, L);
			carryDrawVerts = lastIsDrawLast ? 0 : nDraw;
			carrySubs = lastIsDrawLast ? 0 : nSubs;
			carryFill = lastIsDrawLast ? 0 : nFill;
			carryStroke = lastIsDrawLast ? 0 : nStroke;
			carrySlow = lastIsDrawLast ? 0 : nSlow;
			carrySpVerts = (lastIsDrawLast || lastIsSubLast) ? 0 : nSp;
			carrySpExists = (lastIsDrawLast || lastIsSubLast) ? 0 : nHeadExists;
			dcur = wave_bcast_u64(d, L); // draw of the last command: the next window (if needed) starts here
		}
	}
}


// ------------------------------------------------------------------------------------------------
// k_flatten_build -- single-pass flatten used by vgx_tessellate (steady state).
//
// The two-pass kernel above runs every subdivision twice (count -> scan over draws -> emit) because the ordered
// polyline of the flatten API needs global offsets first. Inside vgx_tessellate the polyline is scratch: only a
// sub-path's vertices must be contiguous. So this kernel subdivides ONCE:
//   - each lane keeps the first VGX_LEAF_SLOTS leaves of its command in LDS (98 % of real-world cubics fit); after the
//     chunk's prefix scan it copies them to the polyline HEAP (transformPos2D applied on the way); only lanes with more
//     leaves re-run their subdivision writing straight to memory;
//   - a wave owns a contiguous run of segments (whole draws) and places their vertices back to back in wave-private
//     blocks of a bump-allocated heap (one atomic per VGX_BUILD_BLOCK vertices, none per chunk); when a chunk does not
//     fit the current block the wave continues in a fresh block, moving along the vertices of the one sub-path that
//     spans into the chunk (only sub-paths must be contiguous);
//   - per sub-path it records {first vertex, count, closed} sparsely at the command-instance index of the sub-path's
//     last command; k_flatten_gather (after the scan over draws has produced ordered mesh indices) turns those into
//     mesh descriptors in the reference's call order.
// Degenerate / serial draws are flagged exactly as in the two-pass kernel and handled by k_flatten_serial.
// ------------------------------------------------------------------------------------------------
// POOL: the cubics of a chunk are subdivided by the whole wave together (pool_sweep, vgx_walk.h) instead of one cubic per
// lane in lock-step; same LDS footprint (the pool's task LIFO takes the place of the per-lane stack + leaf slots).
// Leaves of the per-lane walk kept in LDS per lane before they spill to the wave's global staging area (L2-resident:
// written and read back by the same wave within one chunk). Fewer LDS bytes per wave = more resident waves.
#ifndef VGX_BUILD_LEAF_SLOTS
#define VGX_BUILD_LEAF_SLOTS 8
#endif
#define BLS VGX_BUILD_LEAF_SLOTS
#ifndef VGX_BUILD_UNROLL
#define VGX_BUILD_UNROLL 1 /* the slot leaves of a lane requested together, in front of its stores (0: one LDS round trip per leaf) */
#endif
#ifndef VGX_BUILD_PREFETCH
#define VGX_BUILD_PREFETCH 0 /* 1: the next chunk's records requested in front of this chunk's stores (see `decode`): measured, not kept */
#endif
// -DVGX_BUILD_PROFILE: shader-clock ticks per phase of k_flatten_build summed over all waves into VgxTotals::prof (vgx_get_failure_info)
#ifdef VGX_BUILD_PROFILE
#define FB_CLK() ((unsigned long long)clock64())
#define FB_ACC(i, v) (fbProf[i] += (v))
#else
#define FB_CLK() 0ull
#define FB_ACC(i, v) ((void)(v))
#endif
#ifndef VGX_BUILD_OCC
#define VGX_BUILD_OCC
#endif
struct BuildDec // what a lane of k_flatten_build knows about its command instance before the walk
{
	uint64_t d;            // draw
	const vgx_draw* dr;
	uint64_t subBase;      // sub_prefix[d]
	VgxCmdRec rec;
	float mloc[6];         // the draw's matrix
	float scale, tol;
	uint32_t fillFlags, strokeFlags, serialStatic;
	bool drawHead;
};

template<bool POOL>
__global__ __launch_bounds__(VGX_WAVE) VGX_BUILD_OCC void k_flatten_build(VgxFlattenArgs A)
{
	__shared__ __attribute__((aligned(16))) unsigned char s_mem[POOL ? ((VGX_LDS_LEVELS * 3 + BLS) * VGX_WAVE * 8 > VGX_POOL_BYTES ? (VGX_LDS_LEVELS * 3 + BLS) * VGX_WAVE * 8 : VGX_POOL_BYTES) : (VGX_LDS_LEVELS * 3 + BLS) * VGX_WAVE * 8];
	float2* s_stack = (float2*)s_mem;                            // per-lane walk: pending stack, then the leaf slots
	float2* s_leaf = s_stack + VGX_LDS_LEVELS * 3 * VGX_WAVE;
	const PoolLds pool = pool_carve(s_mem);                      // pooled walk: the same memory
	const int lane = threadIdx.x;
	LdsStack stack;
	stack.base = &s_stack[lane];

	const VgxPathSetDev& ps = A.ps;
	const uint64_t totalCmds = A.cmd_prefix[A.ndraws];
	if (A.totals->status != VGX_OK) { return; }
	if (A.inst_order != nullptr || (A.inst_period != 0 && A.totals->inst_mismatch == 0)) { return; } // instanced batch: k_flatten_inst builds it
	if (A.thin_static && totalCmds <= A.caps.poly_vertices) { return; } // k_flatten_thin built it (this launch is its fallback: see there)
	const uint64_t segItems = vgx_segment_items(totalCmds, gridDim.x);
	const uint64_t numSegments = (totalCmds + segItems - 1) / segItems;
	const uint64_t segsPerWave = (numSegments + gridDim.x - 1) / gridDim.x;
	const uint64_t seg0 = (uint64_t)blockIdx.x * segsPerWave;
	const uint64_t seg1 = (seg0 + segsPerWave < numSegments) ? seg0 + segsPerWave : numSegments;
	if (seg0 >= seg1) {
		return;
	}
	uint64_t dNext = lower_bound_u64(A.cmd_prefix, 0, A.ndraws, seg0 * segItems);
	uint64_t wbase = dNext;
	DrawWindow W = draw_window_load(A, wbase, lane);
	uint64_t blockCur = 0, blockEnd = 0; // wave-private heap block [blockCur, blockEnd)
#ifdef VGX_BUILD_PROFILE
	unsigned long long fbProf[6] = {0, 0, 0, 0, 0, 0};
	const unsigned long long fbT0 = FB_CLK();
#endif

	for (uint64_t seg = seg0; seg < seg1; ++seg) {
		const uint64_t d0 = dNext;
		const uint64_t d1 = advance_lower_bound(A.cmd_prefix, d0, A.ndraws, (seg + 1) * segItems, lane);
		dNext = d1;
		if (d0 == d1) {
			continue;
		}
		const uint64_t C0 = A.cmd_prefix[d0];
		const uint64_t C1 = A.cmd_prefix[d1];

		{
			uint64_t cur = blockCur; // next free heap vertex
			uint64_t dcur = d0;
			int carryDrawVerts = 0, carrySpVerts = 0, carrySubs = 0, carryFill = 0, carryStroke = 0, carrySlow = 0;

			// A chunk's records: window + owner search, command record, the draw's fields. Run for chunk k + 1 BEFORE chunk k's vertices are
			// stored (round 6): loads and stores share one in-order counter, so records requested behind the stores waited until the last of
			// them had reached L2 -- every chunk began with a drained memory pipeline. Requested in front of them they arrive first, and the
			// stores drain under the next walk.
			auto decode = [&](uint64_t chunk, uint64_t dcurAt) {
				const uint64_t ci = chunk + lane;
				const bool valid = ci < C1;
				const uint64_t lastKey = chunk + (VGX_WAVE - 1);
				if (!(wave_bcast_u64(W.prefix, VGX_WAVE - 1) > lastKey) || dcurAt < wbase) {
					wbase = dcurAt;
					W = draw_window_load(A, wbase, lane);
				}
				const bool windowCovers = wave_bcast_u64(W.prefix, VGX_WAVE - 1) > lastKey;
				// owner = last window entry <= my key, searched on 32-bit offsets relative to the chunk start
				const uint32_t wrel = window_rel(W.prefix, chunk);
				const int ownerOfs = window_owner_rel(wrel, valid ? (uint32_t)lane : 0u);
				const uint32_t orel = (uint32_t)__shfl((int)wrel, ownerOfs);
				const int firstOwner = __popcll(wave_ballot(W.prefix <= chunk)) - 1; // draw that owns the chunk's first command
				uint64_t ownerBase = orel > 0 ? chunk + orel : wave_bcast_u64(W.prefix, firstOwner < 0 ? 0 : firstOwner);
				uint32_t pc0 = (uint32_t)__shfl((int)(W.pc0 | ((W.serial & 1u) << 31)), ownerOfs);
				uint32_t serialStatic = pc0 >> 31;
				BuildDec D;
				D.d = d0; D.drawHead = false; D.scale = 1.0f; D.tol = 0.25f; D.fillFlags = 0; D.strokeFlags = 0; D.dr = A.draws; D.subBase = 0;
				D.mloc[0] = 1.0f; D.mloc[1] = 0.0f; D.mloc[2] = 0.0f; D.mloc[3] = 1.0f; D.mloc[4] = 0.0f; D.mloc[5] = 0.0f;
				pc0 &= 0x7FFFFFFFu;
				uint32_t thinPath = ((uint32_t)__shfl((int)W.serial, ownerOfs) >> 1) & 1u; // a moveTo / lineTo / close path: 16-byte thin records
				D.rec.type = VGX_CMD_CLOSE; D.rec.flags = 0; D.rec.na = 0; D.rec.arg_off = 0; D.rec.start[0] = 0.0f; D.rec.start[1] = 0.0f;
				for (int i = 0; i < 8; ++i) { D.rec.a[i] = 0.0f; }
				if (valid) {
					if (windowCovers) {
						D.d = wbase + (uint64_t)ownerOfs;
					} else {
						D.d = find_owner_u64(A.cmd_prefix, d0, d1, ci);
						ownerBase = A.cmd_prefix[D.d];
						const uint32_t path = A.draws[D.d].path;
						pc0 = ps.path_cmd_begin[path];
						serialStatic = ps.path_flags[path] & VGX_PF_SERIAL;
						thinPath = (ps.path_flags[path] >> 1) & 1u;
					}
					D.dr = A.draws + D.d;
					const uint32_t k = (uint32_t)(ci - ownerBase);
					if (thinPath) {
						// my record, the one in front (its point = my start point) and, in front of a CLOSE, the one behind (the
						// sub-path's first point): neighbours of a 1 KB run the wave reads anyway
						const VgxCmdThin t0 = ps.cmdthin[pc0 + k];
						const VgxCmdThin tp = ps.cmdthin[(long long)(pc0 + k) - 1]; // (command 0 of the set reads the padding record; a path's first command never uses it)
						D.rec.type = t0.meta & 0xFFu; D.rec.flags = (t0.meta >> 8) & 0xFFu;
						D.rec.na = D.rec.type == VGX_CMD_CLOSE ? 0u : 2u;
						D.rec.start[0] = tp.x; D.rec.start[1] = tp.y;
						if (D.rec.type == VGX_CMD_CLOSE) { D.rec.a[6] = t0.x; D.rec.a[7] = t0.y; }
						else {
							D.rec.a[0] = t0.x; D.rec.a[1] = t0.y;
							if (D.rec.flags & VGX_CF_NEXT_IS_CLOSE) { const VgxCmdThin tn = ps.cmdthin[pc0 + k + 1]; D.rec.a[6] = tn.x; D.rec.a[7] = tn.y; }
						}
					} else {
						D.rec = ps.cmdrec[pc0 + k];
					}
					D.drawHead = (k == 0);
					D.scale = D.dr->scale; D.tol = D.dr->tess_tol;
					D.fillFlags = D.dr->fill_flags; D.strokeFlags = D.dr->stroke_flags;
					// requested here, used by the placement: behind the walk they cost nothing; issued there they wait behind the stores
					// in front of them (one counter for loads and stores, in order)
#pragma unroll
					for (int i = 0; i < 6; ++i) { D.mloc[i] = D.dr->mtx[i]; }
					D.subBase = A.sub_prefix[D.d];
				}
				D.serialStatic = serialStatic;
				return D;
			};
			BuildDec DN; // the next chunk's records (valid when haveNext)
			bool haveNext = false;
			for (uint64_t chunk = C0; chunk < C1; chunk += VGX_WAVE) {
				const uint64_t ci = chunk + lane;
				const bool valid = ci < C1;
				const unsigned long long fc0 = FB_CLK();
				const BuildDec DC = haveNext ? DN : decode(chunk, dcur);
				const uint64_t d = DC.d;
				const VgxCmdRec& rec = DC.rec;
				const uint32_t type = rec.type, cflags = rec.flags, na = rec.na;
				const bool drawHead = DC.drawHead, drawLast = (cflags & VGX_CF_LAST_IN_PATH) != 0;
				const float scale = DC.scale, tol = DC.tol;
				const uint32_t fillFlags = DC.fillFlags, strokeFlags = DC.strokeFlags;
				const vgx_draw* dr = DC.dr;
				const float* mloc = DC.mloc;
				const uint64_t subBase = DC.subBase;
				const uint32_t serialStatic = DC.serialStatic;
				const bool serialDraw = serialStatic != 0;
				const float* a = rec.a;
				const float* pa = ps.args + rec.arg_off;
				const float* mtx = dr->mtx;
				const V2 start = v2(rec.start[0], rec.start[1]);

#ifdef VGX_BUILD_PROFILE
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the records are here
#endif
				const unsigned long long fc1 = FB_CLK();
				FB_ACC(0, fc1 - fc0); // window + owner search + draw and command records
				// ---- subdivide ONCE: count, detect degenerate cases, keep the first leaves in LDS -------
				int cnt = 0;
				bool slow = false, exists = false, closedHere = false;
				float c1x = a[0], c1y = a[1], c2x = a[2], c2y = a[3], ex = a[4], ey = a[5];
				const bool isCubic = valid && !serialDraw && (type == VGX_CMD_CUBIC_TO || type == VGX_CMD_QUAD_TO);
				if (isCubic && type == VGX_CMD_QUAD_TO) {
					ex = a[2]; ey = a[3];
					vgx_quad_to_cubic(start.x, start.y, a[0], a[1], ex, ey, &c1x, &c1y, &c2x, &c2y);
				}
				const float tessTol = tol / (scale * scale);
				v2f q1, q2, q3, q4;
				q1.x = start.x; q1.y = start.y; q2.x = c1x; q2.y = c1y; q3.x = c2x; q3.y = c2y; q4.x = ex; q4.y = ey;
				bool poolDeep = false; // POOL: my cubic left the pooled path (deeper than VGX_POOL_MAXD / lists full)
				uint32_t poolMask = 0;
				int poolLeaves = 0;
				if (POOL) {
					const uint64_t rootMask = wave_ballot(isCubic);
					if (rootMask) {
						poolLeaves = pool_walk(pool, lane, rootMask, q1, q2, q3, q4, tessTol);
						if (isCubic) {
							const uint32_t fl = pool.flags[lane];
							poolDeep = (fl & VGX_POOL_F_DEEP) != 0;
							poolMask = pool.mask[lane];
							cnt = __popc(poolMask);
							slow = (fl & VGX_POOL_F_SLOW) != 0;
						}
					}
				}
				if (valid && !serialDraw) {
					switch (type) {
					case VGX_CMD_MOVE_TO: cnt = 1; exists = true; break;
					case VGX_CMD_LINE_TO: cnt = 1; slow = v2near(start, v2(a[0], a[1])); break;
					case VGX_CMD_CUBIC_TO:
					case VGX_CMD_QUAD_TO: {
						if (POOL) {
							if (poolDeep) { // count with the per-lane walk; the emit below walks it again, straight to memory
								FastCubicSink<false, false> sink;
								sink.prev = start; sink.n = 0; sink.slow = false;
								wave_flatten_cubic(start.x, start.y, c1x, c1y, c2x, c2y, ex, ey, tessTol, stack, sink);
								sink.flush();
								cnt = (int)sink.n;
								slow = sink.slow;
							}
						} else {
							float2* over = (float2*)A.leaf_overflow + (size_t)blockIdx.x * VGX_BUILD_OVERFLOW * VGX_WAVE + lane;
							uint32_t nLeaves = 0;
							if (!build_flatten_hot<VGX_LDS_LEVELS>(q1, q2, q3, q4, tessTol, &s_stack[lane], &s_leaf[lane], over, &nLeaves, &slow)) {
								BuildCubicSink sink; // nests deeper than the LDS levels: full-depth walk from the root
								sink.prev = start; sink.n = 0; sink.slow = false; sink.slots = &s_leaf[lane]; sink.over = over;
								vgx_flatten_cubic(start.x, start.y, c1x, c1y, c2x, c2y, ex, ey, tessTol, stack, sink);
								nLeaves = sink.n;
								slow = sink.slow;
							}
							cnt = (int)nLeaves;
						}
					} break;
					case VGX_CMD_POLYLINE: {
						const uint32_t npts = na >> 1;
						cnt = (int)npts - ((npts > 0 && v2near(start, v2(pa[0], pa[1]))) ? 1 : 0);
						slow = cnt == 0;
					} break;
					default: break;
					}
				}
				const int rawCnt = cnt;
				const unsigned long long fc2 = FB_CLK();
				FB_ACC(1, fc2 - fc1); // the walk

				// ---- segmented bookkeeping (same as the count pass of k_flatten) ---------------------------
				const uint64_t drawHeads = wave_ballot(valid && drawHead);
				const uint64_t subHeads = wave_ballot(valid && (cflags & VGX_CF_STARTS_SUB));
				const int dh = seg_head(drawHeads, lane);
				const int sh = seg_head(subHeads, lane);
				{
					const int incl1 = wave_incl_scan(cnt, lane);
					const int spBefore1 = seg_rel(incl1 - cnt, sh, carrySpVerts);
					if (valid && !serialDraw && type == VGX_CMD_CLOSE && spBefore1 > 2) { // pathClose, path.cpp:707-726
						closedHere = true;
						if (v2near(start, v2(rec.a[6], rec.a[7]))) { cnt = -1; } // pop: the previous vertex is removed
					}
				}
				const int incl = wave_incl_scan(cnt, lane);
				const int excl = incl - cnt;
				const int inDrawBefore = seg_rel(excl, dh, carryDrawVerts);
				const int spBefore = seg_rel(excl, sh, carrySpVerts);
				const int spTotal = spBefore + cnt;
				const uint64_t existMask = wave_ballot(valid && exists);
				const uint64_t mine = seg_mask_upto(dh, lane);
				const int subsIncl = __popcll(existMask & mine) + (dh < 0 ? carrySubs : 0);
				const bool lastInSub = valid && !serialDraw && (cflags & VGX_CF_LAST_IN_SUB);
				const uint64_t fillMask = wave_ballot(lastInSub && (fillFlags & VGX_FILL_ENABLE) && spTotal >= 3);
				const uint64_t strokeMask = wave_ballot(lastInSub && (strokeFlags & VGX_STROKE_ENABLE) && spTotal >= 2);
				const int fillIncl = __popcll(fillMask & mine) + (dh < 0 ? carryFill : 0);
				const int strokeIncl = __popcll(strokeMask & mine) + (dh < 0 ? carryStroke : 0);
				const uint64_t slowMask = wave_ballot(valid && slow);
				const bool slowDraw = ((slowMask & mine) != 0) || (dh < 0 && carrySlow);

				const int nvalid = (int)((C1 - chunk) < (uint64_t)VGX_WAVE ? (C1 - chunk) : (uint64_t)VGX_WAVE);
				const int L = nvalid - 1;
				const int chunkTotal = wave_bcast(incl, L);
				const unsigned long long fc3 = FB_CLK();
				FB_ACC(2, fc3 - fc2); // scans + bookkeeping
				if (cur + (uint64_t)(chunkTotal > 0 ? chunkTotal : 0) > blockEnd) {
					// The chunk does not fit the wave's block: continue in a fresh one. Only a sub-path's vertices must be
					// contiguous, so the vertices the sub-path that spans into this chunk already has are moved along;
					// the new block has room for twice that prefix, so a very long sub-path is moved O(1) times per
					// vertex.
					const uint64_t carry = (uint64_t)(carrySpVerts > 0 ? carrySpVerts : 0);
					const uint64_t need = carry + (uint64_t)(chunkTotal > 0 ? chunkTotal : 0);
					const uint64_t want = need + carry > (uint64_t)VGX_BUILD_BLOCK ? need + carry : (uint64_t)VGX_BUILD_BLOCK; // doubles the spanning sub-path only
					unsigned long long base = 0;
					if (lane == 0) { base = atomicAdd(&A.totals->poly_heap_cursor, (unsigned long long)want); }
					base = wave_bcast_u64(base, 0);
					if (base + want > A.caps.poly_vertices) {
						if (lane == 0) { atomicCAS(&A.totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_NOSPACE); }
						return;
					}
					if (carry > 0) {
						__threadfence_block(); // the prefix was written by other lanes of this wave
						const float2* src = (const float2*)A.poly + (cur - carry);
						float2* dst = (float2*)A.poly + base;
						for (uint64_t i = (uint64_t)lane; i < carry; i += VGX_WAVE) { dst[i] = src[i]; }
					}
					cur = base + carry;
					blockEnd = base + want;
				}
				// the next chunk's records, requested in front of this chunk's stores
				const uint64_t dcurNext = wave_bcast_u64(d, L); // draw of the last command: the next window (if needed) starts here
				haveNext = VGX_BUILD_PREFETCH && chunk + VGX_WAVE < C1;
				if (haveNext) { DN = decode(chunk + VGX_WAVE, dcurNext); }
				{
					const uint64_t g = cur + (uint64_t)excl; // heap index of my first vertex
					// my last vertex is the one pathClose removes (same decision the CLOSE lane takes)
					uint32_t limit = (valid && !serialDraw) ? (uint32_t)(rawCnt < 0 ? 0 : rawCnt) : 0u;
					if ((cflags & VGX_CF_NEXT_IS_CLOSE) && limit > 0 && spTotal > 2) {
						const V2 endp = (type == VGX_CMD_POLYLINE) ? v2(pa[na - 2], pa[na - 1]) : (type == VGX_CMD_CUBIC_TO ? v2(a[4], a[5]) : (type == VGX_CMD_QUAD_TO ? v2(a[2], a[3]) : v2(a[0], a[1])));
						if (v2near(endp, v2(rec.a[6], rec.a[7]))) { --limit; }
					}
					if (POOL && poolLeaves > 0) { // the listed leaves go straight to their places in the heap
						PoolOutGlobal o;
						o.p = (float2*)A.poly + cur - 64; // excl is -1 at most (a pathClose pop at the chunk's start)
						pool_place(pool, lane, poolLeaves, poolMask, isCubic && !poolDeep, (uint32_t)(excl + 64), limit, mtx, o);
					}
					if (valid && !serialDraw) {
						float* out = A.poly + 2 * g;
						if (type == VGX_CMD_MOVE_TO || type == VGX_CMD_LINE_TO) {
							if (limit > 0) {
								const V2 p = v2xform(v2(a[0], a[1]), mloc);
								*(float2*)out = make_float2(p.x, p.y);
							}
						} else if (type == VGX_CMD_CUBIC_TO || type == VGX_CMD_QUAD_TO) {
							if (POOL) {
								if (poolDeep) {
									FastCubicSink<true, true> sink;
									sink.prev = start; sink.n = 0; sink.slow = false; sink.out = out; sink.writeLimit = limit; sink.mtx = mtx; sink.begin();
									wave_flatten_cubic(start.x, start.y, c1x, c1y, c2x, c2y, ex, ey, tol / (scale * scale), stack, sink);
									sink.flush();
								}
							} else if ((uint32_t)rawCnt <= VGX_LEAF_SLOTS + VGX_BUILD_OVERFLOW) {
								const uint32_t nl = limit < VGX_LEAF_SLOTS ? limit : VGX_LEAF_SLOTS;
#if VGX_BUILD_UNROLL
								float2 lq[VGX_LEAF_SLOTS]; // every slot requested before the first store (one LDS round trip instead of one per leaf)
#pragma unroll
								for (uint32_t i = 0; i < VGX_LEAF_SLOTS; ++i) { lq[i] = s_leaf[i * VGX_WAVE + lane]; }
#pragma unroll
								for (uint32_t i = 0; i < VGX_LEAF_SLOTS; ++i) {
									if (i < nl) { const V2 p = v2xform(v2(lq[i].x, lq[i].y), mloc); *(float2*)(out + 2 * i) = make_float2(p.x, p.y); }
								}
#else
								for (uint32_t i = 0; i < nl; ++i) {
									const float2 q = s_leaf[i * VGX_WAVE + lane];
									const V2 p = v2xform(v2(q.x, q.y), mloc);
									*(float2*)(out + 2 * i) = make_float2(p.x, p.y);
								}
#endif
								const float2* ov = (const float2*)A.leaf_overflow + (size_t)blockIdx.x * VGX_BUILD_OVERFLOW * VGX_WAVE + lane;
								for (uint32_t i = VGX_LEAF_SLOTS; i < limit; ++i) { // same lane wrote these during its subdivision
									const float2 q = ov[(i - VGX_LEAF_SLOTS) * VGX_WAVE];
									const V2 p = v2xform(v2(q.x, q.y), mloc);
									*(float2*)(out + 2 * i) = make_float2(p.x, p.y);
								}
							} else { // more leaves than slots + overflow area: subdivide again, straight to memory
								FastCubicSink<true, true> sink;
								sink.prev = start; sink.n = 0; sink.slow = false; sink.out = out; sink.writeLimit = limit; sink.mtx = mtx; sink.begin();
								wave_flatten_cubic(start.x, start.y, c1x, c1y, c2x, c2y, ex, ey, tol / (scale * scale), stack, sink);
								sink.flush();
							}
						} else if (type == VGX_CMD_POLYLINE && limit < VGX_WAVE) { // (longer ones: the whole wave, below)
							const uint32_t skip = (na >> 1) - (uint32_t)rawCnt;
							for (uint32_t i = 0; i < limit; ++i) {
								const V2 p = v2xform(v2(pa[2 * (i + skip)], pa[2 * (i + skip) + 1]), mloc);
								*(float2*)(out + 2 * i) = make_float2(p.x, p.y);
							}
						}
						if (lastInSub) { // sub-path record, consumed by k_flatten_gather
							VgxSubRec sr;
							sr.first = g - (uint64_t)spBefore; sr.info = (uint32_t)spTotal | (closedHere ? 0x80000000u : 0u); sr.pad = 0;
							A.sub_rec[subBase + (uint64_t)(subsIncl - 1)] = sr; // dense, in draw order: record j of draw d at sub_prefix[d] + j
						}
					}
					{
						// Long POLYLINE commands (pathPolyline copies its points verbatim behind the one epsilon test on the first,
						// path.cpp:684-705): the wave moves them together, 64 consecutive points per step -- coalesced 512-byte loads
						// and stores instead of one lane walking a thousand points while 63 wait (10k polylines x 1k points: one lane
						// per command is 157 waves on the whole chip).
						uint64_t longMask = wave_ballot(valid && !serialDraw && type == VGX_CMD_POLYLINE && limit >= (uint32_t)VGX_WAVE);
						while (longMask) {
							const int src = __builtin_ctzll(longMask);
							longMask &= longMask - 1;
							const uint64_t gS = wave_bcast_u64(g, src);
							const float* paS = (const float*)wave_bcast_u64((uint64_t)(pa + 2 * ((na >> 1) - (uint32_t)rawCnt)), src);
							const float* mS = (const float*)wave_bcast_u64((uint64_t)mtx, src);
							const uint32_t limS = (uint32_t)wave_bcast((int)limit, src);
							const float m0 = mS[0], m1 = mS[1], m2 = mS[2], m3 = mS[3], m4 = mS[4], m5 = mS[5];
							float2* outS = (float2*)A.poly + gS;
							for (uint32_t i = (uint32_t)lane; i < limS; i += VGX_WAVE) {
								const float2 q = *(const float2*)(paS + 2 * (size_t)i);
								outS[i] = make_float2(m0 * q.x + m2 * q.y + m4, m1 * q.x + m3 * q.y + m5); // transformPos2D, vg_util.h:24-28
							}
						}
					}
					{ // draws the serial kernel has to (re)do: static serial paths and degenerate draws, wave-aggregated append
						const bool toSerial = valid && drawLast && (serialDraw || slowDraw);
						const uint64_t sm = wave_ballot(toSerial);
						if (sm) {
							unsigned long long sbase = 0;
							if (lane == 0) { sbase = atomicAdd(&A.totals->num_serial_list, (unsigned long long)__popcll(sm)); }
							sbase = wave_bcast_u64(sbase, 0);
							if (toSerial) { A.serial_list[sbase + (uint64_t)__popcll(sm & lanemask_lt(lane))] = (uint32_t)d; }
						}
					}
					if (valid && drawLast && !serialDraw) {
						vgx_draw_info di;
						di.first_poly_vertex = g - (uint64_t)inDrawBefore; di.first_subpath = 0; di.first_mesh = 0;
						if (slowDraw) {
							di.num_poly_vertices = 0; di.num_subpaths = 0; di.num_meshes = 0; di.flags = 1u;
						} else {
							di.num_poly_vertices = (uint32_t)(inDrawBefore + cnt);
							di.num_subpaths = (uint32_t)subsIncl;
							di.num_meshes = (uint32_t)(fillIncl + strokeIncl);
							di.flags = ((uint32_t)fillIncl << 1);
						}
						A.dinfo[d] = di;
					}
					cur += (uint64_t)(chunkTotal > 0 ? chunkTotal : 0);
				}
				FB_ACC(3, FB_CLK() - fc3); // block switch + placement (stores issued, not waited for) + records
				FB_ACC(4, 1ull);

				// carries into the next chunk
				const int lastIsDrawLast = wave_bcast((int)drawLast, L);
				const int lastIsSubLast = wave_bcast((int)((cflags & VGX_CF_LAST_IN_SUB) != 0), L);
				const int nDraw = wave_bcast(inDrawBefore + cnt, L);
				const int nSp = wave_bcast(spTotal, L);
				const int nSubs = wave_bcast(subsIncl, L);
				const int nFill = wave_bcast(fillIncl, L);
				const int nStroke = wave_bcast(strokeIncl, L);
				const int nSlow = wave_bcast((int)slowDraw, L);
				carryDrawVerts = lastIsDrawLast ? 0 : nDraw;
				carrySubs = lastIsDrawLast ? 0 : nSubs;
				carryFill = lastIsDrawLast ? 0 : nFill;
				carryStroke = lastIsDrawLast ? 0 : nStroke;
				carrySlow = lastIsDrawLast ? 0 : nSlow;
				carrySpVerts = (lastIsDrawLast || lastIsSubLast) ? 0 : nSp;
				dcur = dcurNext;
			}
			blockCur = cur;
		}
	}
#ifdef VGX_BUILD_PROFILE
	fbProf[5] = FB_CLK() - fbT0;
	if (lane == 0) { for (int i = 0; i < 6; ++i) { atomicAdd(&A.totals->prof[i], fbProf[i]); } atomicAdd(&A.totals->prof[6], 1ull); }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_flatten_thin -- k_flatten_build's job for path sets of MOVE_TO / LINE_TO / CLOSE paths only (vgx_thin.h: every decision such a
// path asks for was taken when the set was created). One lane per command instance, VGX_THIN_ITEMS of them per thread; a workgroup
// owns a contiguous run of chunks of VGX_THIN_THREADS x VGX_THIN_ITEMS command instances: ONE binary search over the draws at its
// start, then per chunk the command prefixes of the next (chunk + 1) draws in LDS (every draw has a command: a chunk's owners are
// among them) and a search in LDS per lane. Vertex v of draw d goes to poly[cmd_prefix[d] + v]; the exact builder's draws
// (degenerate paths) are listed and allocate behind poly_heap_cursor = the batch's command instances.
#define VGX_THIN_THREADS 256
template<int VGX_THIN_ITEMS>
__global__ __launch_bounds__(VGX_THIN_THREADS) void k_flatten_thin(VgxFlattenArgs A)
{
	constexpr uint32_t VGX_THIN_CHUNK = VGX_THIN_THREADS * VGX_THIN_ITEMS;
	__shared__ uint64_t s_pref[VGX_THIN_CHUNK + 1];
	__shared__ uint32_t s_path[VGX_THIN_CHUNK + 1];
	const VgxPathSetDev& ps = A.ps;
	const uint32_t tid = threadIdx.x;
	if (A.totals->status != VGX_OK) { return; }
	if (A.inst_order != nullptr || (A.inst_period != 0 && A.totals->inst_mismatch == 0)) { return; } // instanced batch: k_flatten_inst builds it
	const uint64_t totalCmds = A.cmd_prefix[A.ndraws];
	// This kernel places vertex k of draw d at cmd_prefix[d] + k: one heap vertex per COMMAND. A scratch that was sized by a count on
	// another kind of batch may hold the batch's vertices and still not its commands: k_flatten_build (launched behind this kernel,
	// it exits at once otherwise) builds such a batch from the real vertex counts
	if (totalCmds > A.caps.poly_vertices) { return; }
	if (blockIdx.x == 0 && tid == 0) { atomicAdd(&A.totals->poly_heap_cursor, (unsigned long long)totalCmds); }
	const uint64_t numChunks = (totalCmds + VGX_THIN_CHUNK - 1) / VGX_THIN_CHUNK;
	const uint64_t per = (numChunks + gridDim.x - 1) / gridDim.x;
	const uint64_t ch0 = (uint64_t)blockIdx.x * per;
	const uint64_t ch1 = ch0 + per < numChunks ? ch0 + per : numChunks;
	if (ch0 >= ch1) { return; }
	// the draw that owns the run's first command instance: the last d with cmd_prefix[d] <= key. A 256-ary search by the whole
	// workgroup (two or three dependent loads for any batch instead of log2(ndraws)): cmd_prefix[lo] <= key < cmd_prefix[hi] throughout
	uint64_t dcur;
	{
		const uint64_t key = ch0 * VGX_THIN_CHUNK;
		uint64_t lo = 0, hi = A.ndraws;
		while (hi - lo > 1) {
			const uint64_t step = (hi - lo + VGX_THIN_THREADS - 1) / VGX_THIN_THREADS;
			uint64_t idx = lo + (uint64_t)(tid + 1) * step;
			if (idx > hi) { idx = hi; }
			const uint32_t cnt = (uint32_t)__syncthreads_count(A.cmd_prefix[idx] <= key ? 1 : 0); // the samples do not decrease with tid: the first cnt are <= key (never all: the last one is hi's)
			uint64_t nhi = lo + (uint64_t)(cnt + 1) * step;
			if (nhi > hi) { nhi = hi; }
			lo = lo + (uint64_t)cnt * step;
			hi = nhi;
		}
		dcur = lo;
	}
	for (uint64_t ch = ch0; ch < ch1; ++ch) {
		const uint64_t c0 = ch * VGX_THIN_CHUNK;
		__syncthreads(); // (the previous chunk's searches are done)
		// The window: command prefix and path of draws dcur, dcur + 1, ... as far as the chunk reaches -- 256 entries at a time, counting
		// the entries that begin at or before the NEXT chunk's first command instance (they are sorted: a prefix of the window). A
		// block with fewer than 256 of them ends the window (long paths: one load per thread); entries behind it are never read.
		const uint64_t keyNext = c0 + VGX_THIN_CHUNK;
		uint32_t nextOwn = 0; // entries 1 .. nextOwn begin at or before keyNext: owner of the next chunk's first command instance (uniform)
		if (tid == 0) { s_pref[0] = A.cmd_prefix[dcur]; s_path[0] = A.draws[dcur].path; }
		for (uint32_t b0 = 0; b0 < VGX_THIN_CHUNK; b0 += VGX_THIN_THREADS) {
			const uint32_t i = b0 + 1u + tid;
			const uint64_t idx = dcur + i;
			const uint64_t v = idx <= A.ndraws ? A.cmd_prefix[idx] : ~0ull;
			s_pref[i] = v;
			s_path[i] = idx < A.ndraws ? A.draws[idx].path : 0u;
			const uint32_t c = (uint32_t)__syncthreads_count(v <= keyNext ? 1 : 0); // (a barrier: the entries written so far are visible)
			nextOwn += c;
			if (c < VGX_THIN_THREADS) { break; }
		}
		uint32_t own[VGX_THIN_ITEMS];
		uint64_t base[VGX_THIN_ITEMS];
		bool valid[VGX_THIN_ITEMS];
#pragma unroll
		for (int it = 0; it < VGX_THIN_ITEMS; ++it) {
			const uint64_t ci = c0 + (uint64_t)it * VGX_THIN_THREADS + tid;
			valid[it] = ci < totalCmds;
			const uint64_t key = valid[it] ? ci : c0;
			uint32_t lo = 0, hi = nextOwn; // s_pref[0] <= c0 <= key < keyNext: the owner is among entries 0 .. nextOwn
			while (lo < hi) {
				const uint32_t mid = (lo + hi + 1) >> 1;
				if (s_pref[mid] <= key) { lo = mid; } else { hi = mid - 1; }
			}
			own[it] = lo; base[it] = s_pref[lo];
		}
		// the draws' words and the paths' records (the path index came with the window), then the commands' records: all items' loads
		// of a stage issued together
		uint32_t path[VGX_THIN_ITEMS], ff[VGX_THIN_ITEMS], sf[VGX_THIN_ITEMS];
		float m[VGX_THIN_ITEMS][6];
#pragma unroll
		for (int it = 0; it < VGX_THIN_ITEMS; ++it) {
			const vgx_draw* dr = A.draws + (dcur + own[it]);
			path[it] = s_path[own[it]]; ff[it] = dr->fill_flags; sf[it] = dr->stroke_flags;
#pragma unroll
			for (int j = 0; j < 6; ++j) { m[it][j] = dr->mtx[j]; }
		}
		VgxThinPath q[VGX_THIN_ITEMS];
#pragma unroll
		for (int it = 0; it < VGX_THIN_ITEMS; ++it) { q[it] = ps.thin_path[path[it]]; }
		VgxCmdThin t[VGX_THIN_ITEMS];
#pragma unroll
		for (int it = 0; it < VGX_THIN_ITEMS; ++it) {
			const uint64_t ci = valid[it] ? c0 + (uint64_t)it * VGX_THIN_THREADS + tid : base[it];
			t[it] = ps.cmdthin[q[it].pc0 + (uint32_t)(ci - base[it])];
		}
#pragma unroll
		for (int it = 0; it < VGX_THIN_ITEMS; ++it) {
			if (valid[it]) {
				const uint64_t d = dcur + own[it];
				if (vgx_thin_lane(q[it], t[it], ps.thin_sub, m[it], ff[it], sf[it], base[it], A.sub_prefix + d, A.poly, A.sub_rec, A.dinfo + d)) {
					const unsigned long long at = atomicAdd(&A.totals->num_serial_list, 1ull);
					A.serial_list[at] = (uint32_t)d;
				}
			}
		}
		dcur += nextOwn;
	}
}

// One lane per draw, after the scan over draws: turns the sparse sub-path records of k_flatten_build into mesh
// descriptors (+ closed-form mesh-table sizes) at their ORDERED indices: fill meshes by sub-path, then stroke meshes
// (the reference's call order, vg.cpp:3099-3131 then 3448-3485). Serial draws were written by k_flatten_serial.
__device__ __forceinline__ void flatten_gather_body(const VgxFlattenArgs& A, uint64_t tid, uint64_t nthreads)
{
	const VgxPathSetDev& ps = A.ps;
	const uint64_t P = A.inst_period;
	const bool periodic = A.inst_order == nullptr && P != 0 && A.totals->inst_mismatch == 0u; // k_flatten_inst built the batch, OpCmdPrefix::apply_size
	for (uint64_t d = tid; d < A.ndraws; d += nthreads) {
		const vgx_draw_info di = A.dinfo[d];
		if ((di.flags & 1u) || di.num_meshes == 0) { continue; }
		// Everything the loop reads is loaded BEFORE its first store (the draw record as a local copy, the first sub-path
		// records into registers): a load issued after a store waits for that store as well (vmcnt counts both, in order),
		// and the compiler cannot move it up past a store that may alias.
		const vgx_draw drLocal = A.draws[d];
		const vgx_draw* dr = &drLocal;
		const uint32_t path = dr->path;
		const uint32_t sb0 = ps.path_sub_begin[path], sb1 = ps.path_sub_begin[path + 1];
		// both flatten kernels store record j of draw d at sub_prefix[d] + j; periodic batches keep the first period's prefixes only
		const uint64_t cbase = periodic ? (d / P) * A.sub_prefix[P] + A.sub_prefix[d % P] : A.sub_prefix[d];
		const uint32_t fillFlags = dr->fill_flags, strokeFlags = dr->stroke_flags;
		const uint32_t numFill = di.flags >> 1;
		uint32_t f = 0, s = 0, subIndex = 0;
		VgxSubRec pre[4];
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) { if (sb0 + j < sb1) { pre[j] = A.sub_rec[cbase + j]; } }
		for (uint32_t sb = sb0; sb < sb1; ++sb) {
			const uint64_t ci = cbase + subIndex;
			VgxSubRec sr;
			if (subIndex == 0) { sr = pre[0]; } else if (subIndex == 1) { sr = pre[1]; } else if (subIndex == 2) { sr = pre[2]; } else if (subIndex == 3) { sr = pre[3]; }
			else { sr = A.sub_rec[ci]; }
			const uint32_t info = sr.info;
			const uint32_t n = info & 0x7FFFFFFFu;
			const bool closed = (info >> 31) != 0;
			const uint64_t first = sr.first;
			if ((fillFlags & VGX_FILL_ENABLE) && n >= 3) {
				vgx_write_mesh(A.mdesc, A.mtab, di.first_mesh + f, dr, (uint32_t)d, subIndex, (fillFlags & VGX_FILL_AA) ? VGX_MESH_FILL_AA : VGX_MESH_FILL, closed, first, n, A.mprep, A.poly, sr.pad);
				++f;
			}
			if ((strokeFlags & VGX_STROKE_ENABLE) && n >= 2) {
				const uint32_t kd = !(strokeFlags & VGX_STROKE_AA) ? VGX_MESH_STROKE : ((strokeFlags & VGX_STROKE_THIN) ? VGX_MESH_STROKE_AA_THIN : VGX_MESH_STROKE_AA);
				if (vgx_write_mesh(A.mdesc, A.mtab, di.first_mesh + numFill + s, dr, (uint32_t)d, subIndex, kd, closed, first, n, A.mprep, A.poly)) {
					atomicAdd(&A.totals->num_round_meshes, 1u);
				}
				++s;
			}
			++subIndex;
		}
	}
}

__global__ __launch_bounds__(256) void k_flatten_gather(VgxFlattenArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	flatten_gather_body(A, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// The same behind k_flat1 (vgx_tessellate's one-walk route, round 6): the draws' records are complete and ordered already (places from the
// look-back), the sub-path records are the flatten API's vgx_subpath at the draw's first_subpath; checks the batch's meshes against the
// scratch the last count sized (k_flat1 knows the caller's polyline / sub-path capacities only).
__global__ __launch_bounds__(256) void k_flatten_gather_ordered(VgxFlattenArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	if (A.totals->sizes.num_meshes > A.caps.meshes) { // (uniform: every thread reads the same words)
		if (blockIdx.x == 0 && threadIdx.x == 0) { atomicCAS(&A.totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_NOSPACE); }
		return;
	}
	for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < A.ndraws; d += (uint64_t)gridDim.x * blockDim.x) {
		const vgx_draw_info di = A.dinfo[d];
		if ((di.flags & 1u) || di.num_meshes == 0) { continue; } // (serial draws: k_flatten_serial wrote their meshes)
		const vgx_draw drLocal = A.draws[d]; // (loaded in front of the first store: see flatten_gather_body)
		const vgx_draw* dr = &drLocal;
		const uint32_t fillFlags = dr->fill_flags, strokeFlags = dr->stroke_flags;
		const uint32_t numFill = di.flags >> 1;
		uint32_t f = 0, sk = 0;
		vgx_subpath pre[4];
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) { if (j < di.num_subpaths) { pre[j] = A.subs[di.first_subpath + j]; } }
		for (uint32_t j = 0; j < di.num_subpaths; ++j) {
			vgx_subpath r;
			if (j == 0) { r = pre[0]; } else if (j == 1) { r = pre[1]; } else if (j == 2) { r = pre[2]; } else if (j == 3) { r = pre[3]; }
			else { r = A.subs[di.first_subpath + j]; }
			const uint32_t n = r.num_vertices;
			const bool closed = (r.flags & 1u) != 0;
			if ((fillFlags & VGX_FILL_ENABLE) && n >= 3) {
				vgx_write_mesh(A.mdesc, A.mtab, di.first_mesh + f, dr, (uint32_t)d, j, (fillFlags & VGX_FILL_AA) ? VGX_MESH_FILL_AA : VGX_MESH_FILL, closed, r.first_vertex, n, A.mprep, A.poly, 0u);
				++f;
			}
			if ((strokeFlags & VGX_STROKE_ENABLE) && n >= 2) {
				const uint32_t kd = !(strokeFlags & VGX_STROKE_AA) ? VGX_MESH_STROKE : ((strokeFlags & VGX_STROKE_THIN) ? VGX_MESH_STROKE_AA_THIN : VGX_MESH_STROKE_AA);
				if (vgx_write_mesh(A.mdesc, A.mtab, di.first_mesh + numFill + sk, dr, (uint32_t)d, j, kd, closed, r.first_vertex, n, A.mprep, A.poly)) {
					atomicAdd(&A.totals->num_round_meshes, 1u);
				}
				++sk;
			}
		}
	}
}

// ---- private (per-lane) pending stack for the serial kernel ---------------------------------------------
struct PrivStack
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

// k_flatten_serial: ONE LANE PER DRAW runs the exact sequential builder (PathSim = vg::Path semantics) for
//   - paths with closed shapes / arcs (statically flagged at upload: their vertices depend on trigonometry
//     recurrences and, for arcs, on computed end points), and
//   - draws the lane-parallel kernel flagged as degenerate (epsilon de-dup hit, dropped subdivision piece).
// Slow by construction, exact by construction; everything else never enters this kernel.
template<bool EMIT, bool XFORM>
__device__ __forceinline__ void flatten_serial_body(const VgxFlattenArgs& A, uint64_t tid, uint64_t nthreads)
{
	const VgxPathSetDev& ps = A.ps;
	PrivStack stack;
	// BUILD mode: k_flatten_build listed the draws to do; otherwise every draw is inspected
	const uint64_t nwork = A.build_mode ? (uint64_t)A.totals->num_serial_list : A.ndraws;
	for (uint64_t w = tid; w < nwork; w += nthreads) {
		const uint64_t d = A.build_mode ? (uint64_t)A.serial_list[w] : w;
		const vgx_draw* dr = A.draws + d;
		const uint32_t path = dr->path;
		const bool serial = (ps.path_flags[path] & VGX_PF_SERIAL) || (A.dinfo[d].flags & 1u);
		const uint32_t pc0 = ps.path_cmd_begin[path], pc1 = ps.path_cmd_begin[path + 1];
		if (!serial || pc0 == pc1) { continue; }
		PathSim<EMIT, XFORM> sim;
		sim.scale = dr->scale; sim.tol = dr->tess_tol; sim.mtx = dr->mtx; sim.poly = A.poly;
		sim.drawIndex = (uint32_t)d; sim.fillFlags = dr->fill_flags; sim.strokeFlags = dr->stroke_flags; sim.draw = dr;
		if (!EMIT) {
			sim.polyBase = 0; sim.subs = nullptr; sim.subBase = 0; sim.mdesc = nullptr; sim.mprep = nullptr; sim.mtab = nullptr; sim.meshBase = 0;
			sim.numFillTotal = 0; sim.limit = 0;
			sim.init();
			sim.run(ps, pc0, pc1, stack);
			vgx_draw_info di;
			di.first_poly_vertex = 0; di.first_subpath = 0; di.first_mesh = 0;
			di.num_poly_vertices = sim.nverts; di.num_subpaths = sim.nsubs; di.num_meshes = sim.nfill + sim.nstroke;
			di.flags = 1u | (sim.nfill << 1);
			if (A.build_mode) { // the polyline heap of the single-pass path: exact allocation for this draw
				di.first_poly_vertex = atomicAdd(&A.totals->poly_heap_cursor, (unsigned long long)sim.nverts);
				if (di.first_poly_vertex + sim.nverts > A.caps.poly_vertices) { atomicCAS(&A.totals->status, (uint32_t)VGX_OK, (uint32_t)VGX_E_NOSPACE); }
			}
			A.dinfo[d] = di;
		} else {
			const vgx_draw_info di = A.dinfo[d];
			sim.polyBase = di.first_poly_vertex; sim.subs = A.subs; sim.subBase = di.first_subpath;
			sim.mdesc = A.mdesc; sim.mprep = A.mprep; sim.mtab = A.mtab; sim.meshBase = di.first_mesh; sim.numFillTotal = di.flags >> 1;
			sim.limit = di.num_poly_vertices;
			sim.init();
			sim.run(ps, pc0, pc1, stack);
			if (sim.numRound) { atomicAdd(&A.totals->num_round_meshes, sim.numRound); }
		}
	}
}

template<bool EMIT, bool XFORM>
__global__ __launch_bounds__(256) void k_flatten_serial(VgxFlattenArgs A)
{
	if (A.totals->status != VGX_OK) { return; }
	flatten_serial_body<EMIT, XFORM>(A, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// ------------------------------------------------------------------------------------------------
// Frame-sized batches (at most VGX_SMALL_DRAWS draws: one vg-renderer frame is a few hundred paths) are bound by the
// number of DEPENDENT launches, ~7 us each on this platform whether launched or graph-replayed: 17 of them made a
// 240-path drawing cost 125 us. Everything between the lane-parallel flatten and the element kernels needs device-wide
// results but only a few thousand items, so ONE workgroup does it with block barriers instead of kernel boundaries:
//   k_small_front   zero the totals + per-draw records, scan of command instances per draw
//   k_flatten_build (unchanged)
//   k_small_middle  exact serial builder for the listed draws (count), scan over the draws, mesh descriptors (gather +
//                   serial emit), Round-join mesh sizes, scan over the meshes (+ the caller's mesh table), totals published
//   k_fill, k_stroke (unchanged)
// Same device functions as the large-batch kernels, so the results are identical by construction.
// Values another thread produced with an ATOMIC (status word, heap cursor, Round-mesh counter) are read with agent-scope
// atomic loads: atomics execute in L2 and do not refresh this CU's L1 line.
// ------------------------------------------------------------------------------------------------
#define VGX_SMALL_THREADS 1024

__device__ __forceinline__ uint32_t small_status(const VgxTotals* t) { return __hip_atomic_load(&t->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(VGX_SMALL_THREADS) void k_small_front(OpCmdPrefix op, vgx_draw_info* dinfo)
{
	__shared__ Sum3 s_wave[VGX_SMALL_THREADS / 64];
	{ // the zeroing the large-batch path does with two memsets
		uint32_t* t = (uint32_t*)op.totals;
		for (uint32_t i = threadIdx.x; i < sizeof(VgxTotals) / 4; i += VGX_SMALL_THREADS) { t[i] = 0u; }
		uint32_t* d = (uint32_t*)dinfo;
		for (uint64_t i = threadIdx.x; i < op.ndraws * (sizeof(vgx_draw_info) / 4); i += VGX_SMALL_THREADS) { d[i] = 0u; }
	}
	__syncthreads();
	block_scan_all<OpCmdPrefix, VGX_SMALL_THREADS>(op, s_wave);
}

struct VgxSmallArgs
{
	VgxFlattenArgs F;   // build_mode = 1, mprep set
	VgxStrokeArgs S;    // draws / poly / mdesc / mprep / mtab / totals for the Round-join sizing
	OpDrawInfo opDraws;
	OpMeshAll opMeshes;
	vgx_sizes* dev_sizes;   // may be null
	uint32_t* dev_status;   // may be null
};

__global__ __launch_bounds__(VGX_SMALL_THREADS) void k_small_middle(VgxSmallArgs K)
{
	__shared__ Sum3 s_wave[VGX_SMALL_THREADS / 64];
	VgxTotals* T = K.F.totals;
	const uint64_t tid = threadIdx.x;
	// (1) draws the lane-parallel kernel handed to the exact serial builder: count + heap allocation
	if (small_status(T) == VGX_OK) { flatten_serial_body<false, false>(K.F, tid, VGX_SMALL_THREADS); }
	__syncthreads();
	// (2) scan over the draws (reads status through size(): plain load of a word only atomics write -> use the fresh value)
	{
		OpDrawInfo op = K.opDraws;
		if (small_status(T) != VGX_OK) { op.ndraws = 0; }
		block_scan_all<OpDrawInfo, VGX_SMALL_THREADS>(op, s_wave);
	}
	// (3) mesh descriptors + per-mesh constants: sub-path records of the lane-parallel kernel, then the serial draws' emit pass
	if (small_status(T) == VGX_OK) {
		flatten_gather_body(K.F, tid, VGX_SMALL_THREADS);
		flatten_serial_body<true, true>(K.F, tid, VGX_SMALL_THREADS);
	}
	__syncthreads();
	// (4) meshes with Round joins: one wave per mesh
	const bool ok4 = small_status(T) == VGX_OK;
	const uint64_t numMeshes = ok4 ? T->sizes.num_meshes : 0;
	if (ok4 && __hip_atomic_load(&T->num_round_meshes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
		const int lane = threadIdx.x & 63;
		for (uint64_t mi = threadIdx.x >> 6; mi < numMeshes; mi += VGX_SMALL_THREADS / 64) {
			if (K.S.mtab[mi].num_vertices != VGX_MESH_NEEDS_COUNT) { continue; } // wave-uniform
			uint32_t sv, si;
			round_mesh_size(make_mesh_ctx(K.S.mdesc[mi], K.S.mprep[mi], K.S.draws, 0, K.S.poly), lane, &sv, &si);
			if (lane == 0) { K.S.mtab[mi].num_vertices = sv; K.S.mtab[mi].num_indices = si; }
		}
	}
	__syncthreads();
	// (5) scan over the meshes: element / vertex / index offsets, the caller's mesh table, totals
	{
		OpMeshAll op = K.opMeshes;
		// the status word may have been set by an atomic in (3) / (4): a plain load could differ between waves (stale L1 line)
		op.fixedSize = 1;
		op.fixedCount = (small_status(T) == VGX_OK) ? T->sizes.num_meshes : 0;
		block_scan_all<OpMeshAll, VGX_SMALL_THREADS>(op, s_wave);
	}
	// (6) publish
	if (threadIdx.x == 0) {
		if (K.dev_sizes) { *K.dev_sizes = T->sizes; }
		if (K.dev_status) { *K.dev_status = small_status(T); }
	}
}

} // namespace

void vgx_launch_small_front(const void* opCmdPrefix, vgx_draw_info* dinfo, hipStream_t s)
{
	hipLaunchKernelGGL(k_small_front, dim3(1), dim3(VGX_SMALL_THREADS), 0, s, *(const OpCmdPrefix*)opCmdPrefix, dinfo);
}

void vgx_launch_small_middle(const VgxFlattenArgs& f, const VgxStrokeArgs& st, const void* opDraws, const void* opMeshes, vgx_sizes* devSizes, uint32_t* devStatus, hipStream_t s)
{
	VgxSmallArgs k;
	k.F = f; k.S = st; k.opDraws = *(const OpDrawInfo*)opDraws; k.opMeshes = *(const OpMeshAll*)opMeshes; k.dev_sizes = devSizes; k.dev_status = devStatus;
	hipLaunchKernelGGL(k_small_middle, dim3(1), dim3(VGX_SMALL_THREADS), 0, s, k);
}

void vgx_launch_flatten_build(const VgxFlattenArgs& a, int waves, hipStream_t s, bool serialCount)
{
	if (a.inst_period || a.inst_order) { vgx_launch_flatten_inst(a, a.inst_waves, s); } // one of the two exits at once (device-side check)
	// waves < VGX_BUILD_WAVES is a testing knob (VGX_BUILD_WAVES in the environment at vgx_create): a handful of waves makes
	// small batches run through the heap's block switches and sub-path moves that otherwise need > 8192 vertices per wave
	if (a.thin_static) { // a set of MOVE_TO / LINE_TO / CLOSE paths: the static layout (vgx_thin.h)
		const dim3 grid(a.ndraws <= VGX_SMALL_DRAWS ? 64 : 2048); // (frame-sized batches: workgroups that find no chunk still cost their first loads)
		if (a.thin_static == 2) { hipLaunchKernelGGL(k_flatten_thin<2>, grid, dim3(VGX_THIN_THREADS), 0, s, a); } // (VGX_THIN_STATIC=2: two command instances per thread)
		else { hipLaunchKernelGGL(k_flatten_thin<4>, grid, dim3(VGX_THIN_THREADS), 0, s, a); }
		hipLaunchKernelGGL(k_flatten_build<false>, dim3(a.ndraws <= VGX_SMALL_DRAWS ? 64 : waves), dim3(VGX_WAVE), 0, s, a); // (exits at once unless the scratch holds fewer vertices than the batch has commands)
	} else if (a.pool_walk) {
		hipLaunchKernelGGL(k_flatten_build<true>, dim3(waves), dim3(VGX_WAVE), 0, s, a);
	} else {
		hipLaunchKernelGGL(k_flatten_build<false>, dim3(waves), dim3(VGX_WAVE), 0, s, a);
	}
	if (serialCount) { hipLaunchKernelGGL((k_flatten_serial<false, false>), dim3(1024), dim3(256), 0, s, a); } // count + heap allocation
}

void vgx_launch_flatten_gather_ordered(const VgxFlattenArgs& a, hipStream_t s) // behind k_flat1 (+ its k_flatten_serial emit)
{
	hipLaunchKernelGGL(k_flatten_gather_ordered, dim3(2048), dim3(256), 0, s, a);
}

void vgx_launch_flatten_gather(const VgxFlattenArgs& a, hipStream_t s)
{
	hipLaunchKernelGGL(k_flatten_gather, dim3(2048), dim3(256), 0, s, a);
	hipLaunchKernelGGL((k_flatten_serial<true, true>), dim3(1024), dim3(256), 0, s, a);
}

// k_flatten_serial alone (vgx_flatten's one-walk kernel, vgx_flat1.hip, does the lane-parallel part): count = every draw of a
// statically serial path; emit = the draws flagged in dinfo (a.build_mode: only the listed ones)
void vgx_launch_flatten_serial(bool emit, const VgxFlattenArgs& a, hipStream_t s)
{
	const int sb = a.build_mode ? 64 : 1024;
	if (!emit) { hipLaunchKernelGGL((k_flatten_serial<false, false>), dim3(sb), dim3(256), 0, s, a); }
	else if (a.apply_transform) { hipLaunchKernelGGL((k_flatten_serial<true, true>), dim3(sb), dim3(256), 0, s, a); }
	else { hipLaunchKernelGGL((k_flatten_serial<true, false>), dim3(sb), dim3(256), 0, s, a); }
}

void vgx_launch_flatten(bool emit, const VgxFlattenArgs& a, int numBlocks, hipStream_t s)
{
	// lane-parallel kernel first (it flags degenerate draws), then the one-lane-per-draw exact kernel
	const int sb = 1024;
	if (!emit) {
		hipLaunchKernelGGL((k_flatten<false, false>), dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
		hipLaunchKernelGGL((k_flatten_serial<false, false>), dim3(sb), dim3(256), 0, s, a);
	} else if (a.apply_transform) {
		hipLaunchKernelGGL((k_flatten<true, true>), dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
		hipLaunchKernelGGL((k_flatten_serial<true, true>), dim3(sb), dim3(256), 0, s, a);
	} else {
		hipLaunchKernelGGL((k_flatten<true, false>), dim3(numBlocks), dim3(VGX_WAVE), 0, s, a);
		hipLaunchKernelGGL((k_flatten_serial<true, false>), dim3(sb), dim3(256), 0, s, a);
	}
}
