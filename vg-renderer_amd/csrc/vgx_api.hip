// vgx_api.hip -- implementation of the C-ABI declared in include/vgx.h.
//
// Host side of the batch geometry path: context (device scratch, grow-only), path-set validation and
// upload, the four device-wide scans and the launch sequence
//   cmd-prefix scan -> flatten<count> -> draw scan -> flatten<emit> -> element-prefix scan
//   -> stroke<count> -> mesh scan -> stroke<emit>
// All lengths that are results of earlier kernels stay in device memory (VgxTotals); the asynchronous
// entry point vgx_tessellate never synchronises with the host.
#include "vgx_internal.h"
#include "vgx_scan.h"
#include "vgx_scan_ops.h"
#include "vgx_pathsim.h"
#include "vgx_inst.h"
#include "vgx_flat1.h"
#include "vgx_thin.h"
#include "vgx_pathset_dev.h"
#include "vgx_tile.h"
#include "vgx_mscan.h"
#include <vector>
#include <atomic>
#include <unordered_map>
#include <string.h>
#include <math.h>
#include <new>
#include <stdlib.h>
#include <time.h>
#include <stdio.h>


static int vgxGridBlocks()
{
	static int g = 0;
	if (!g) { const char* e = getenv("VGX_GRID_BLOCKS"); g = e ? atoi(e) : 32768; if (g < 256) { g = 256; } }
	return g;
}
#define VGX_GRID_BLOCKS vgxGridBlocks() // one-wave workgroups, each owning a contiguous run of segments (>> resident waves: no tail)

// Grid of the element kernels for a batch whose element count is at most `maxElements` (a bound the host knows: the
// caller's vertex capacity -- every element emits at least one vertex). Frame-sized batches should not pay for
// dispatching 32768 workgroups that exit at once (~8 us per kernel).
static int vgxElementGrid(uint64_t maxElements)
{
	const uint64_t want = maxElements / (2 * 64) + 1; // >= 2 chunks per wave
	const uint64_t full = (uint64_t)VGX_GRID_BLOCKS;
	return (int)(want < 256 ? 256 : (want > full ? full : want));
}

struct vgx_pathset
{
	VgxPathSetDev dev;
	void* blob; // single device allocation holding every array
	size_t blobBytes;
	uint32_t maxCmdsPerPath;
	bool hasSerial; // some path takes the exact serial builder (ARC / ARC_TO / closed shapes)
	bool hasEmpty;  // some path has no commands
	uint32_t numSubs; // sub-paths of the set (entries of sub_last_cmd / thin_sub)
	bool thinStatic; // every path is MOVE_TO / LINE_TO / CLOSE only and the static layout tables are filled (vgx_thin.h): k_flatten_thin builds its batches
	uint64_t gen; // unique per vgx_pathset_create (process-wide counter): identifies the path set where an address could be reused
};

#define VGX_PS_POOL 8
#define VGX_PS_POOL_MAX ((size_t)8 << 20)
struct DevBuf
{
	void* p;
	size_t cap;
};

struct vgx_ctx
{
	int device;
	int lastHipError;
	int pendingHipError; // first failure of an asynchronous helper call (memset) inside the current entry point
	// grow-only device scratch
	DevBuf asmJump0, asmJump1, asmStart, meshBase; // draw-command assembly scratch (only when armed)
	vgx_assembly asmCfg;
	bool asmArmed;
	DevBuf subPrefix; // exclusive scan of the draws' static sub-path counts
	DevBuf cmdPrefix, cmdCnt, subFirst, leafOverflow, serialList, dinfo, poly, subs, mdesc, elemPrefix, elemPrefixS, mprep, mtab, partial, totals;
	DevBuf tileTab;                      // k_emit_tiles (vgx_tile.hip): the tile table of the current call
	bool tileHint;                       // the last ordinary vgx_tessellate_count saw fills and closed Miter AA / Thin strokes only (the tile kernel's batches)
	uint64_t optBigEmitMin;              // vertex capacity from which a call launches the tile kernel / k_stroke_long (2^18; VGX_BIG_EMIT_MIN: testing knob, 0 = every call)
	int optStrokeLong;                   // VGX_STROKE_LONG=0: batches of long polylines through k_stroke like any other (no LDS-staged stores)
	int optTileEmit;                     // VGX_TILE_EMIT=0: ordinary batches through k_fill + k_stroke_simple as before round 6
	DevBuf psTemp;                       // vgx_pathset_create: temporaries of the device-side build (vgx_pathset.hip)
	hipStream_t psStream;                // ... its stream (created at the first call)
	struct VgxPsTotals* hostPs;          // ... pinned: what the build reports
	void* psImage; size_t psImageCap;    // ... pinned: the raw part of a frame-sized set, assembled here and uploaded in one copy
	DevBuf psPool[VGX_PS_POOL];          // ... blobs of dropped frame-sized sets, recycled (hipFree synchronises)
	void* psStage[2]; hipEvent_t psStageEv[2]; bool psStageBusy[2]; uint64_t psStageK; int optPsStage, optPsNoSmall; // VGX_PS_UPLOAD=stage: own pinned staging
	DevBuf f1SegDraw, f1Segs;            // vgx_flatten (vgx_flat1.hip): segment table, look-back records
	int optF1Waves, optF1Cap, optF1Seg;  // its grid (persistent one-wave workgroups), the leaf-list capacity of the kernel instance and the segment bucket (0 = chosen per batch)
	// what the last vgx_flatten call produced (copied to pinned memory behind the call, read by the next call WITHOUT waiting for
	// it): polyline vertices per command instance decide how many commands a segment may hold before its leaves overflow the list
	unsigned long long* hostF1;          // pinned: [0] vertices, [1] command instances, [2] tag of the batch they belong to
	uint64_t f1Tag;
	DevBuf gatherSizes;                  // vgx_gather_sizes: [nranks][4] uint64
	DevBuf partBounds;                   // vgx_partition: [nparts + 1] bounds + [nparts] weights
	struct VgxRccl* rccl;                // RCCL entry points, bound at the first vgx_gather* call
	// options, read from the environment ONCE at vgx_create (tuning / testing knobs)
	int optTwoPass, optBuildWaves, optPoolWalk, optNoSmall, optConcurrentEmit;
	int optTessFlat1;                    // VGX_TESS_FLAT1: 1 (default) = vgx_tessellate flattens batches of long curves with the one-walk kernel (k_flat1), 0 = never, 2 = every eligible batch
	// vgx_tessellate's one-walk route, decided and sized by the last vgx_tessellate_count (f1Route*): the path set it is for, the kernel
	// instance / bucket size, the segments the look-back tables hold
	bool f1Route; const vgx_pathset* f1RoutePs; uint64_t f1RoutePsGen; int f1RouteCap; uint32_t f1RouteSegMax; uint64_t f1RouteSegBound;
	int optThinStatic;                   // VGX_THIN_STATIC=0: lineTo-only path sets through k_flatten_build like any other (default: k_flatten_thin, vgx_thin.h)
	int optInst, optInstWaves; uint32_t optInstBlock; // instanced flatten kernel (vgx_inst.hip): on / grid / lane block
	int optInstPerm;                                  // periodic batches of different scales: permute instances (1, default) or sort draws by (path, class) (0)
	uint32_t optInstClasses;                          // grouped mode: tolerance classes per path when the instances differ in scale (VGX_INST_CLASSES)
	// instanced batches: period of the path sequence found by the last vgx_tessellate_count (0 = none). vgx_tessellate
	// re-checks it on the device for the draws it is given.
	uint32_t instPeriod;
	// ... or, when the draws reuse paths without repeating one sequence (at least 32 draws per used path on average): 1 =
	// grouped mode, every vgx_tessellate sorts the draws by path first (k_inst_hist / k_inst_plan / k_inst_scatter)
	int instGrouped;
	// grouped mode, instances of different scales: the sort key is (path, tolerance class) with this many classes per path
	// (1 = by path only), so that the lanes of a wave flatten with nearly the same tolerance and stay in lock-step
	uint32_t instClasses;
	DevBuf instHist, instCursor, instKeyStart, instStart, instTaskStart, instTaskPath, instOrder;
	// periodic mode, instances of different scales: every vgx_tessellate sorts the INSTANCES by tolerance class (vgx_launch_inst_perm)
	int instPermOn;
	DevBuf instPerm, instPermHist;
	uint64_t instCapPaths, instCapKeys, instCapTasks, instCapDraws;
	// template mode (vgx_tmpl.hip): the first period of an instanced batch whose instances differ in transform / colours only,
	// flattened once in local space by the last vgx_tessellate_count
	int optTmpl; uint32_t optTmplTile;
	uint32_t tmplTileSize;               // elements per tile of the current template
	DevBuf tmplTile;                     // [tiles] VgxTmplTile
	bool tmplOn;
	const vgx_pathset* tmplPs;
	uint64_t tmplPsGen;                  // generation id of tmplPs when the template was built (an address can be reused by a later path set)
	uint32_t tmplPeriod;
	vgx_sizes tmplInst;                  // sizes of one instance
	DevBuf tmplPoly, tmplMesh, tmplMtab, tmplElem, tmplDraws;
	// ... in several flavours ("classes", e.g. the same drawing at a few scales): one template over the concatenated class
	// representatives, a per-instance table of output places, a per-workgroup table (instance, tile)
	int optTmplClasses;
	int optTmplRound;                    // VGX_TMPL_ROUND=0: batches with Round joins keep the ordinary pipeline
	int optTmplBatch;                    // vgx_set_static_batches / VGX_TMPL_BATCH=1: a batch without a period becomes ONE template (the whole draw list = one instance)
	bool tmplIsBatch;                    // the current template is such a batch-wide one
	uint32_t tmplClasses;                // 1: every instance repeats the first period
	uint32_t tmplGeneral;                // stroke styles of the template: 0 closed Miter AA / Thin only, 1 + open Miter with Butt / Square caps, 2 + general
	uint64_t tmplNumWg, tmplNDraws;      // several classes: workgroups of one step; the batch size the per-instance table was built for
	vgx_sizes tmplTotal;                 // sizes of the whole batch
	DevBuf tmplHash, tmplInstCls, tmplClsRep, tmplCls, tmplIinfo, tmplWg, tmplClsSum;
	uint32_t tmplRound;                  // Round-join stroke meshes per instance (tmplGeneral == 3): their sizes, and every place behind them, are counted per step
	uint32_t tmplRoundElems;             // their elements per instance
	uint32_t tmplRoundLds;               // Round-join meshes of the largest class (several classes; else = tmplRound)
	uint64_t tmplRelemWords;             // several classes: table words of the whole batch (sum over the instances of their class's Round-join elements)
	DevBuf tmplTrmesh, tmplTmsz;         // template: the Round-join meshes (mesh, first element among the Round-join elements); per mesh its sizes (VgxTmplArgs::tmsz)
	DevBuf tmplRsz, tmplRelem, tmplMplace, tmplItot, tmplIplace; // the per-step tables of such a template (VgxTmplArgs)
	hipStream_t sideStream; hipEvent_t forkEv, joinEv; // optConcurrentEmit only
	VgxCaps caps; // element capacities matching the buffers above
	uint64_t capDraws;
	VgxTotals* hostTotals; // pinned
	// state of the last *_count call (what _emit continues from)
	const vgx_pathset* lastPs;
	const vgx_draw* lastDraws;
	uint64_t lastNDraws;
	int lastStage; // 0 none, 1 flatten counted, 2 tessellate counted
	// profiling
	int profiling;
	// a ring of event sets, one per profiled call: vgx_get_stage_times reads the last call's, vgx_get_stage_times_avg the average over
	// the last calls (a caller that times K back-to-back calls gets per-kernel durations of THOSE calls, without a sync in between)
	hipEvent_t ev[VGX_PROF_RING][VGX_MAX_STAGES + 1];
	const char* evName[VGX_PROF_RING][VGX_MAX_STAGES];
	uint32_t numEv[VGX_PROF_RING];
	uint64_t profCalls; // profiled calls since profiling was switched on
	uint32_t evSet;     // set of the current call
	bool evCreated;
};

static void vgx_rccl_release(vgx_ctx* ctx);

namespace {


#define HIPCHK(ctx, call)                                 \
	do {                                                  \
		hipError_t e_ = (call);                           \
		if (e_ != hipSuccess) {                           \
			(ctx)->lastHipError = (int)e_;                \
			return VGX_E_HIP;                             \
		}                                                 \
	} while (0)

int ensure(vgx_ctx* ctx, DevBuf& b, size_t bytes)
{
	if (bytes <= b.cap) {
		return VGX_OK;
	}
	// grow with 12.5 % head room so that near-identical batches do not reallocate
	size_t want = bytes + bytes / 8 + 256;
	void* fresh = nullptr;
	hipError_t e = hipMalloc(&fresh, want);
	if (e != hipSuccess && b.p) { // not enough room for old + new at once: release the old block first (contents are scratch)
		(void)hipFree(b.p);
		b.p = nullptr;
		b.cap = 0;
		e = hipMalloc(&fresh, want);
	}
	if (e != hipSuccess) {
		ctx->lastHipError = (int)e;
		return VGX_E_HIP;
	}
	if (b.p) { (void)hipFree(b.p); }
	b.p = fresh;
	b.cap = want;
	return VGX_OK;
}

// Every entry point that takes a context runs on the context's device, whatever device is current on the calling
// thread (a torch rank, another context), and leaves the caller's current device as it found it.
struct DeviceGuard
{
	int prev;
	bool switched;
	explicit DeviceGuard(const vgx_ctx* ctx) : prev(-1), switched(false)
	{
		if (ctx && hipGetDevice(&prev) == hipSuccess && prev != ctx->device) { switched = hipSetDevice(ctx->device) == hipSuccess; }
	}
	~DeviceGuard() { if (switched) { (void)hipSetDevice(prev); } }
};

// Launch failures (bad configuration, device lost) surface as hipGetLastError: every entry point that enqueues kernels
// ends with this instead of pretending success.
int launchStatus(vgx_ctx* ctx)
{
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		ctx->lastHipError = (int)e;
		return VGX_E_HIP;
	}
	if (ctx->pendingHipError) { // an asynchronous memset of this call failed
		ctx->lastHipError = ctx->pendingHipError;
		ctx->pendingHipError = 0;
		return VGX_E_HIP;
	}
	return VGX_OK;
}

void noteHip(vgx_ctx* ctx, hipError_t e)
{
	if (e != hipSuccess && !ctx->pendingHipError) { ctx->pendingHipError = (int)e; }
}

void mark(vgx_ctx* ctx, hipStream_t s, const char* name)
{
	const uint32_t k = ctx->evSet;
	if (!ctx->profiling || ctx->numEv[k] >= VGX_MAX_STAGES) {
		return;
	}
	ctx->evName[k][ctx->numEv[k]] = name;
	++ctx->numEv[k];
	(void)hipEventRecord(ctx->ev[k][ctx->numEv[k]], s);
}

void markBegin(vgx_ctx* ctx, hipStream_t s)
{
	if (!ctx->profiling) {
		return;
	}
	if (!ctx->evCreated) {
		for (int r = 0; r < VGX_PROF_RING; ++r) { for (int i = 0; i <= VGX_MAX_STAGES; ++i) { (void)hipEventCreate(&ctx->ev[r][i]); } }
		ctx->evCreated = true;
	}
	ctx->evSet = (uint32_t)(ctx->profCalls % VGX_PROF_RING);
	++ctx->profCalls;
	ctx->numEv[ctx->evSet] = 0;
	(void)hipEventRecord(ctx->ev[ctx->evSet][0], s);
}

struct OpCacheInst // shape cache: vertices / indices / meshes of every instance's mesh range -> output offsets
{
	vgx_cache_desc cache;
	const vgx_cache_instance* inst;
	uint64_t ninst;
	uint64_t* meshPrefix;
	uint64_t* vertPrefix;
	uint64_t* idxPrefix;
	VgxTotals* totals;
	VgxCaps caps;
	__device__ uint64_t size() const { return ninst; }
	__device__ uint64_t cv(uint64_t k) const { return k < cache.num_meshes ? cache.meshes[k].first_vertex : cache.num_vertices; }
	__device__ uint64_t ci(uint64_t k) const { return k < cache.num_meshes ? cache.meshes[k].first_index : cache.num_indices; }
	__device__ Sum3 load(uint64_t i) const
	{
		Sum3 r = sum3_zero();
		const vgx_cache_instance in = inst[i];
		if (in.first_mesh > cache.num_meshes || (uint64_t)in.num_meshes > cache.num_meshes - in.first_mesh) { set_status(totals, VGX_E_INVALID_ARG); return r; }
		r.a = in.num_meshes;
		r.b = cv(in.first_mesh + in.num_meshes) - cv(in.first_mesh);
		r.c = ci(in.first_mesh + in.num_meshes) - ci(in.first_mesh);
		return r;
	}
	__device__ void store(uint64_t i, Sum3 e) const { meshPrefix[i] = e.a; vertPrefix[i] = e.b; idxPrefix[i] = e.c; }
	__device__ void finish(Sum3 t) const
	{
		meshPrefix[ninst] = t.a; vertPrefix[ninst] = t.b; idxPrefix[ninst] = t.c;
		totals->sizes.num_meshes = t.a;
		totals->sizes.num_vertices = t.b;
		totals->sizes.num_indices = t.c;
		if (t.a > caps.meshes || t.b > caps.vertices || t.c > caps.indices) { set_status(totals, VGX_E_NOSPACE); }
	}
};

struct OpConcaveFills // concave fills: vertices / indices of every fill's mesh -> offsets + mesh records
{
	const vgx_contour* contours;
	uint64_t ncontours;
	const vgx_concave_fill* fills;
	uint64_t nfills;
	vgx_mesh* mtab;
	vgx_mesh* meshesOut; // caller's table (may be null)
	VgxTotals* totals;
	VgxCaps caps;
	__device__ uint64_t size() const { return nfills; }
	__device__ bool counts(uint64_t i, uint64_t* nv, uint64_t* ni) const
	{
		const vgx_concave_fill f = fills[i];
		*nv = 0; *ni = 0;
		if (f.first_contour > ncontours || (uint64_t)f.num_contours > ncontours - f.first_contour) { return false; }
		uint64_t cv = 0; // contour vertices of the fill (its contours are stored back to back)
		if (f.num_contours) {
			const vgx_contour c0 = contours[f.first_contour], c1 = contours[f.first_contour + f.num_contours - 1];
			if (c1.first_vertex < c0.first_vertex) { return false; }
			cv = c1.first_vertex + c1.num_vertices - c0.first_vertex;
		}
		*nv = 2 * cv + f.num_tess_vertices;
		*ni = 6 * cv + f.num_tess_indices;
		return true;
	}
	__device__ Sum3 load(uint64_t i) const
	{
		Sum3 r = sum3_zero();
		uint64_t nv, ni;
		if (!counts(i, &nv, &ni)) { set_status(totals, VGX_E_INVALID_ARG); return r; }
		if (nv > 65536u) { set_status(totals, VGX_E_MESH_TOO_LARGE); } // uint16 indices
		r.a = nv; r.b = ni;
		return r;
	}
	__device__ void store(uint64_t i, Sum3 e) const
	{
		uint64_t nv, ni;
		(void)counts(i, &nv, &ni);
		vgx_mesh m;
		m.first_vertex = e.a; m.first_index = e.b;
		m.num_vertices = (uint32_t)nv; m.num_indices = (uint32_t)ni;
		m.draw = (uint32_t)i;
		m.subpath_kind = (uint32_t)VGX_MESH_CONCAVE_FILL_AA << 28;
		mtab[i] = m;
		if (meshesOut && i < caps.meshes) { meshesOut[i] = m; }
	}
	__device__ void finish(Sum3 t) const
	{
		totals->sizes.num_meshes = nfills;
		totals->sizes.num_vertices = t.a;
		totals->sizes.num_indices = t.b;
		if (t.a > caps.vertices || t.b > caps.indices || nfills > caps.meshes) { set_status(totals, VGX_E_NOSPACE); }
	}
};

struct OpSubMeshes // stroker-level entry: one or two meshes per vertex list -> mesh descriptors + closed-form sizes
{
	const vgx_subpath* subs;
	const uint32_t* subDraw;
	const vgx_draw* draws;
	uint64_t nsubs, ndraws;
	VgxMeshDesc* mdesc;
	vgx_mesh* mtab;
	VgxTotals* totals;
	__device__ uint64_t size() const { return nsubs; }
	__device__ Sum3 load(uint64_t i) const
	{
		Sum3 r = sum3_zero();
		const uint32_t di = subDraw[i];
		if (di >= ndraws) { set_status(totals, VGX_E_INVALID_ARG); return r; }
		const vgx_draw* d = draws + di;
		const uint32_t n = subs[i].num_vertices;
		const uint32_t sf = d->stroke_flags;
		if ((sf & VGX_STROKE_ENABLE) && (VGX_STROKE_CAP(sf) > 2u || VGX_STROKE_JOIN(sf) > 2u)) { set_status(totals, VGX_E_INVALID_ARG); return r; }
		r.a = (((d->fill_flags & VGX_FILL_ENABLE) && n >= 3) ? 1u : 0u) + (((sf & VGX_STROKE_ENABLE) && n >= 2) ? 1u : 0u);
		r.b = n;
		return r;
	}
	__device__ void store(uint64_t i, Sum3 e) const
	{
		const uint32_t di = subDraw[i];
		if (di >= ndraws) { return; }
		const vgx_draw* d = draws + di;
		const vgx_subpath sp = subs[i];
		const bool closed = (sp.flags & 1u) != 0;
		uint64_t m = e.a;
		if ((d->fill_flags & VGX_FILL_ENABLE) && sp.num_vertices >= 3) {
			vgx_write_mesh(mdesc, mtab, m, d, di, (uint32_t)i, (d->fill_flags & VGX_FILL_AA) ? VGX_MESH_FILL_AA : VGX_MESH_FILL, closed, sp.first_vertex, sp.num_vertices);
			++m;
		}
		const uint32_t sf = d->stroke_flags;
		if ((sf & VGX_STROKE_ENABLE) && sp.num_vertices >= 2) {
			const uint32_t k = !(sf & VGX_STROKE_AA) ? VGX_MESH_STROKE : ((sf & VGX_STROKE_THIN) ? VGX_MESH_STROKE_AA_THIN : VGX_MESH_STROKE_AA);
			if (vgx_write_mesh(mdesc, mtab, m, d, di, (uint32_t)i, k, closed, sp.first_vertex, sp.num_vertices)) { atomicAdd(&totals->num_round_meshes, 1u); }
		}
	}
	__device__ void finish(Sum3 t) const
	{
		totals->sizes.num_meshes = t.a;
		totals->sizes.num_poly_vertices = t.b;
		totals->sizes.num_subpaths = nsubs;
	}
};

__global__ void k_tmpl_nospace(VgxTotals* t, vgx_sizes need, uint32_t aux)
{
	t->sizes = need;
	t->status = VGX_E_NOSPACE;
	t->fail_reason = VGX_FAIL_OUT_CAPACITY;
	t->fail_aux = aux;
}

__global__ void k_publish(const VgxTotals* t, vgx_sizes* devSizes, uint32_t* devStatus)
{
	if (devSizes) { *devSizes = t->sizes; }
	if (devStatus) { *devStatus = t->status; }
}

// ---- pipeline pieces ----------------------------------------------------------------------------------
VgxFlattenArgs flattenArgs(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, int applyTransform)
{
	VgxFlattenArgs a;
	a.ps = ps->dev;
	a.draws = draws;
	a.ndraws = ndraws;
	a.cmd_prefix = (const uint64_t*)ctx->cmdPrefix.p;
	a.sub_prefix = (const uint64_t*)ctx->subPrefix.p;
	a.cmd_cnt = (uint32_t*)ctx->cmdCnt.p;
	a.dinfo = (vgx_draw_info*)ctx->dinfo.p;
	a.poly = (float*)ctx->poly.p;
	a.subs = (vgx_subpath*)ctx->subs.p;
	a.mdesc = (VgxMeshDesc*)ctx->mdesc.p;
	a.mprep = nullptr;
	a.mtab = (vgx_mesh*)ctx->mtab.p;
	a.totals = (VgxTotals*)ctx->totals.p;
	a.caps = ctx->caps;
	a.apply_transform = applyTransform;
	a.sub_rec = (VgxSubRec*)ctx->subFirst.p;
	a.build_mode = 0;
	a.pool_walk = ctx->optPoolWalk;
	a.thin_static = ps->thinStatic ? ctx->optThinStatic : 0;
	a.leaf_overflow = (float*)ctx->leafOverflow.p;
	a.serial_list = (uint32_t*)ctx->serialList.p;
	a.inst_period = 0; a.inst_block = ctx->optInstBlock; a.inst_waves = ctx->optInstWaves; a.inst_perm = nullptr;
	a.inst_order = nullptr; a.inst_start = nullptr; a.inst_task_start = nullptr; a.inst_task_path = nullptr;
	return a;
}

// Period the instanced kernel may assume for a batch of `ndraws` draws (0: command-parallel kernel). The device checks
// draws[i].path == draws[i % period].path again on every call (OpCmdPrefix).
uint32_t instPeriodFor(const vgx_ctx* ctx, uint64_t ndraws)
{
	const uint32_t P = ctx->instPeriod;
	if (!P || !ctx->optInst || ndraws % P != 0 || ndraws / P < VGX_INST_MIN_INSTANCES) { return 0; }
	return P;
}

// Grouped mode for this call: the count pass chose it and the sort buffers hold a batch of this size over this path set.
bool instGroupedFor(const vgx_ctx* ctx, const vgx_pathset* ps, uint64_t ndraws)
{
	return ctx->instGrouped && ctx->optInst && !instPeriodFor(ctx, ndraws) && ndraws > VGX_SMALL_DRAWS && ndraws <= ctx->instCapDraws
		&& ps->dev.npaths <= ctx->instCapPaths && (uint64_t)ps->dev.npaths * ctx->instClasses <= ctx->instCapKeys && ndraws < 0xFFFFFFFFull;
}

// scratch of the grouped mode: histogram / cursor / ranges per path, task table, draw order
int ensureInstGroup(vgx_ctx* ctx, uint32_t npaths, uint32_t nc, uint64_t ndraws, bool full)
{
	int st;
	const size_t nkeys = (size_t)npaths * nc;
	if ((st = ensure(ctx, ctx->instHist, (nkeys + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->instStart, ((size_t)npaths + 1) * sizeof(uint64_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->instTaskStart, ((size_t)npaths + 1) * sizeof(uint64_t))) != VGX_OK) { return st; }
	if (full) {
		const uint64_t tasks = ndraws / 64 + (uint64_t)npaths + 2; // every used path rounds up once
		if ((st = ensure(ctx, ctx->instCursor, (nkeys + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
		if ((st = ensure(ctx, ctx->instKeyStart, (nkeys + 1) * sizeof(uint64_t))) != VGX_OK) { return st; }
		if ((st = ensure(ctx, ctx->instTaskPath, tasks * sizeof(uint32_t))) != VGX_OK) { return st; }
		if ((st = ensure(ctx, ctx->instOrder, (ndraws + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
		ctx->instCapTasks = ctx->instTaskPath.cap / sizeof(uint32_t);
		ctx->instCapDraws = ctx->instOrder.cap / sizeof(uint32_t) - 1;
		uint64_t ck = ctx->instHist.cap / sizeof(uint32_t) - 1;
		{ const uint64_t c2 = ctx->instCursor.cap / sizeof(uint32_t) - 1, c3 = ctx->instKeyStart.cap / sizeof(uint64_t) - 1; if (c2 < ck) { ck = c2; } if (c3 < ck) { ck = c3; } }
		ctx->instCapKeys = ck;
		uint64_t cp = ctx->instStart.cap / sizeof(uint64_t) - 1;
		{ const uint64_t c4 = ctx->instTaskStart.cap / sizeof(uint64_t) - 1; if (c4 < cp) { cp = c4; } }
		ctx->instCapPaths = cp;
	}
	return VGX_OK;
}

void setInstArgs(vgx_ctx* ctx, const vgx_pathset* ps, uint64_t ndraws, VgxFlattenArgs& a)
{
	a.inst_period = instPeriodFor(ctx, ndraws);
	a.inst_perm = nullptr;
	if (a.inst_period && ctx->instPermOn && ndraws / a.inst_period <= ctx->instPerm.cap / sizeof(uint32_t)) { a.inst_perm = (const uint32_t*)ctx->instPerm.p; }
	if (instGroupedFor(ctx, ps, ndraws)) {
		a.inst_order = (const uint32_t*)ctx->instOrder.p; a.inst_start = (const uint64_t*)ctx->instStart.p;
		a.inst_task_start = (const uint64_t*)ctx->instTaskStart.p; a.inst_task_path = (const uint32_t*)ctx->instTaskPath.p;
	}
}

int ensureDrawBuffers(vgx_ctx* ctx, uint64_t ndraws)
{
	int st;
	if ((st = ensure(ctx, ctx->cmdPrefix, (ndraws + 1) * sizeof(uint64_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->subPrefix, (ndraws + 1) * sizeof(uint64_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->dinfo, (ndraws + 1) * sizeof(vgx_draw_info))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->serialList, (ndraws + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->partial, VGX_SCAN_BLOCKS * sizeof(Sum3))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->totals, sizeof(VgxTotals))) != VGX_OK) { return st; }
	// what the grow-only buffers hold, not the last batch's size
	uint64_t cap = ctx->cmdPrefix.cap / sizeof(uint64_t) - 1;
	{ const uint64_t c2 = ctx->dinfo.cap / sizeof(vgx_draw_info) - 1, c3 = ctx->serialList.cap / sizeof(uint32_t) - 1, c4 = ctx->subPrefix.cap / sizeof(uint64_t) - 1; if (c2 < cap) { cap = c2; } if (c3 < cap) { cap = c3; } if (c4 < cap) { cap = c4; } }
	ctx->capDraws = cap;
	return VGX_OK;
}

int readTotals(vgx_ctx* ctx, hipStream_t s)
{
	HIPCHK(ctx, hipMemcpyAsync(ctx->hostTotals, ctx->totals.p, sizeof(VgxTotals), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	return VGX_OK;
}

// stage 1: command-instance prefix
void runCmdPrefix(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, hipStream_t s, uint32_t instPeriod = 0)
{
	noteHip(ctx, hipMemsetAsync(ctx->totals.p, 0, sizeof(VgxTotals), s));
	OpCmdPrefix op;
	op.draws = draws; op.pathCmdBegin = ps->dev.path_cmd_begin; op.npaths = ps->dev.npaths; op.ndraws = ndraws;
	op.prefix = (uint64_t*)ctx->cmdPrefix.p; op.totals = (VgxTotals*)ctx->totals.p; op.cap = ctx->caps.cmd_instances;
	op.pathSubBegin = ps->dev.path_sub_begin; op.subPrefix = (uint64_t*)ctx->subPrefix.p;
	op.period = instPeriod;
	vgx_device_scan(op, (Sum3*)ctx->partial.p, s, ndraws);
	mark(ctx, s, "scan_cmd_prefix");
}

// stage 2: flatten count + draw scan
void runFlattenCount(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, hipStream_t s)
{
	noteHip(ctx, hipMemsetAsync(ctx->dinfo.p, 0, ndraws * sizeof(vgx_draw_info), s));
	VgxFlattenArgs a = flattenArgs(ctx, ps, draws, ndraws, 1);
	vgx_launch_flatten(false, a, VGX_GRID_BLOCKS, s);
	mark(ctx, s, "flatten_count");
	OpDrawInfo op;
	op.dinfo = (vgx_draw_info*)ctx->dinfo.p; op.ndraws = ndraws; op.totals = (VgxTotals*)ctx->totals.p; op.caps = ctx->caps; op.keepPolyBase = 0;
	vgx_device_scan(op, (Sum3*)ctx->partial.p, s, ndraws);
	mark(ctx, s, "scan_draws");
}

// single-pass flatten of the steady-state entry point: build (subdivide once, polyline -> heap) -> scan -> gather
void runFlattenBuild(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, hipStream_t s)
{
	noteHip(ctx, hipMemsetAsync(ctx->dinfo.p, 0, ndraws * sizeof(vgx_draw_info), s));
	VgxFlattenArgs a = flattenArgs(ctx, ps, draws, ndraws, 1);
	a.build_mode = 1;
	setInstArgs(ctx, ps, ndraws, a); // periodic: the same value runCmdPrefix checked the draws against
	if (a.inst_perm) { // periodic mode, instances of different scales: this batch's instances sorted by tolerance class
		vgx_launch_inst_perm(draws, ndraws / a.inst_period, a.inst_period, ctx->instClasses, (uint32_t*)ctx->instPermHist.p, (uint32_t*)ctx->instPerm.p, (VgxTotals*)ctx->totals.p, ctx->optInstPerm == 2, s);
		mark(ctx, s, "inst_group");
	}
	if (a.inst_order) { // grouped mode: this batch's draws sorted by path
		vgx_launch_inst_group(draws, ndraws, ps->dev.npaths, ctx->instClasses, (uint32_t*)ctx->instHist.p, (uint32_t*)ctx->instCursor.p, (uint64_t*)ctx->instKeyStart.p,
			(uint64_t*)ctx->instStart.p, (uint64_t*)ctx->instTaskStart.p, (uint32_t*)ctx->instTaskPath.p, ctx->instCapTasks, (uint32_t*)ctx->instOrder.p,
			(VgxTotals*)ctx->totals.p, ctx->partial.p, s);
		mark(ctx, s, "inst_group");
	}
	a.mprep = (VgxMeshPrep*)ctx->mprep.p; // k_flatten_gather / k_flatten_serial write the per-mesh constants with the descriptors
	vgx_launch_flatten_build(a, ctx->optBuildWaves, s);
	mark(ctx, s, "flatten_build");
	OpDrawInfo op;
	op.dinfo = (vgx_draw_info*)ctx->dinfo.p; op.ndraws = ndraws; op.totals = (VgxTotals*)ctx->totals.p; op.caps = ctx->caps; op.keepPolyBase = 1;
	vgx_device_scan(op, (Sum3*)ctx->partial.p, s, ndraws);
	mark(ctx, s, "scan_draws");
	vgx_launch_flatten_gather(a, s);
	mark(ctx, s, "flatten_gather");
}

// vgx_tessellate's flatten stage through the ordered one-walk kernel (vgx_flat1.hip; round 6): command prefix -> k_flat1
// (tasks cut at the roots, leaves staged in LDS, places by look-back: the polyline lands dense and in draw order in the scratch, the per-draw and
// sub-path records complete) -> the exact builder's draws -> totals -> mesh descriptors from the ordered records. No heap, no scan over the
// draws. For batches of LONG curves, where k_flatten_build's lanes spill
// their leaves past the LDS slots and walk in lock-step with the deepest cubic of the chunk (the count decides: >= 10 polyline vertices per command instance).
void runFlattenOneWalk(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, hipStream_t s)
{
	{
		const uint64_t savedCmdCap = ctx->caps.cmd_instances; // k_flat1 needs no per-command scratch (see vgx_flatten)
		ctx->caps.cmd_instances = ~0ull;
		runCmdPrefix(ctx, ps, draws, ndraws, s);
		ctx->caps.cmd_instances = savedCmdCap;
	}
	VgxFlattenArgs a = flattenArgs(ctx, ps, draws, ndraws, 1);
	a.mprep = (VgxMeshPrep*)ctx->mprep.p; // k_flatten_serial / the gather write the per-mesh constants with the descriptors
	const uint64_t segBound = ctx->f1RouteSegBound;
	VgxF1Args x;
	x.seg_draw = (uint64_t*)ctx->f1SegDraw.p; x.segs = (VgxF1Seg*)ctx->f1Segs.p; x.grps = x.segs + segBound;
	x.cap_poly = ctx->caps.poly_vertices; x.cap_subs = ctx->caps.subpaths; x.pass = 0; x.read_flags = 0; x.has_empty = ps->hasEmpty ? 1 : 0;
	x.seg_max = ctx->f1RouteSegMax; x.tag = 0;
	if (ps->hasSerial) { // statically serial paths: the exact builder counts their draws first (it marks them in dinfo)
		noteHip(ctx, hipMemsetAsync(ctx->dinfo.p, 0, ndraws * sizeof(vgx_draw_info), s));
		VgxFlattenArgs ac = a;
		ac.mdesc = nullptr; ac.mtab = nullptr; ac.mprep = nullptr;
		vgx_launch_flatten_serial(false, ac, s);
	}
	vgx_launch_flat1(a, x, ctx->optF1Waves ? ctx->optF1Waves : 2048, ctx->f1RouteCap, ps->hasSerial, s);
	VgxFlattenArgs ae = a; // the exact builder's draws: vertices, sub-path records AND mesh descriptors at the places k_flat1 gave them
	ae.build_mode = ps->hasSerial ? 0 : 1;
	vgx_launch_flatten_serial(true, ae, s);
	vgx_launch_flat1_publish(a, x, nullptr, nullptr, s);
	mark(ctx, s, "flatten_one_walk");
	vgx_launch_flatten_gather_ordered(a, s);
	mark(ctx, s, "flatten_gather");
}

// segments the look-back tables of the one-walk route must hold for ANY draw list of `ndraws` draws on this path set
static uint64_t f1RouteSegmentsFor(const vgx_pathset* ps, uint64_t ndraws, uint32_t segMax)
{
	const uint64_t minItems = segMax < 32 ? segMax : 32;
	return ndraws * (uint64_t)(ps->maxCmdsPerPath ? ps->maxCmdsPerPath : 1) / minItems + 2;
}

// prepDone: the flatten stage already wrote the per-mesh constants (single-pass pipeline), no k_mesh_prepare pass
void runStrokeCount(vgx_ctx* ctx, const vgx_draw* draws, const VgxCaps& outCaps, int checkCaps, hipStream_t s, const float* poly = nullptr, bool prepDone = false, vgx_mesh* meshesOut = nullptr)
{
	VgxStrokeArgs a;
	a.draws = draws; a.poly = poly ? poly : (const float*)ctx->poly.p; a.mdesc = (const VgxMeshDesc*)ctx->mdesc.p;
	a.elem_prefix = nullptr; a.elem_prefix_fill = (const uint64_t*)ctx->elemPrefix.p; a.elem_prefix_stroke = (const uint64_t*)ctx->elemPrefixS.p;
	a.mprep = (VgxMeshPrep*)ctx->mprep.p; a.mtab = (vgx_mesh*)ctx->mtab.p;
	a.pos = nullptr; a.color = nullptr; a.idx = nullptr; a.meshes_out = nullptr; a.mesh_base = nullptr;
	a.totals = (VgxTotals*)ctx->totals.p; a.caps = outCaps; a.tile_mode = 0; a.no_long = 0;
	if (!prepDone) { vgx_launch_mesh_prepare(a, s); }
	vgx_launch_stroke(false, a, 32768, s); // k_round_sizes: Round-join mesh sizes (exits immediately without Round joins); one wave per mesh: 10 000 long polylines want more than 4 096 waves
	mark(ctx, s, "mesh_prepare");
	OpMeshAll op;
	op.mdesc = (const VgxMeshDesc*)ctx->mdesc.p; op.mtab = (vgx_mesh*)ctx->mtab.p;
	op.prefixFill = (uint64_t*)ctx->elemPrefix.p; op.prefixStroke = (uint64_t*)ctx->elemPrefixS.p;
	op.totals = (VgxTotals*)ctx->totals.p; op.caps = outCaps; op.checkCaps = checkCaps;
	op.meshesOut = meshesOut; op.fixedSize = 0; op.fixedCount = 0;
	vgx_device_scan(op, (Sum3*)ctx->partial.p, s, ctx->caps.meshes);
	mark(ctx, s, "scan_meshes");
}

// Draw-command assembly (armed by vgx_set_assembly): partition of the mesh sequence into vertex buffers, per-mesh index
// base, draw-command table. Runs between the scan over meshes and the emit kernels, which add the base to every index.
int runAssemble(vgx_ctx* ctx, const vgx_mesh_out* out, hipStream_t s, const vgx_draw* draws = nullptr)
{
	const uint32_t maxVB = ctx->asmCfg.max_vb_vertices ? ctx->asmCfg.max_vb_vertices : 65536u;
	const uint64_t meshCap = ctx->mtab.cap / sizeof(vgx_mesh);
	uint64_t need = 2 * (out->cap_vertices / maxVB) + 4; // two consecutive vertex buffers always hold > maxVB vertices
	if (need > meshCap + 1) { need = meshCap + 1; }
	if (meshCap >= 0xFFFFFFFFull) { return VGX_E_RANGE; } // the jump tables hold 32-bit mesh indices
	uint64_t capStart = 2;
	while (capStart < need) { capStart <<= 1; }
	int st;
	if ((st = ensure(ctx, ctx->asmJump0, (meshCap + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->asmJump1, (meshCap + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->asmStart, capStart * sizeof(uint32_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->meshBase, (meshCap + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
	VgxAsmArgs a;
	a.mtab = (const vgx_mesh*)ctx->mtab.p;
	a.jump0 = (uint32_t*)ctx->asmJump0.p; a.jump1 = (uint32_t*)ctx->asmJump1.p;
	a.start = (uint32_t*)ctx->asmStart.p; a.cap_start = capStart;
	a.mesh_base = (uint32_t*)ctx->meshBase.p;
	a.drawcmds = ctx->asmCfg.drawcmds; a.cap_drawcmds = ctx->asmCfg.cap_drawcmds; a.dev_num_drawcmds = ctx->asmCfg.dev_num_drawcmds;
	a.max_vb = maxVB;
	a.totals = (VgxTotals*)ctx->totals.p;
	a.flags = draws ? ctx->asmCfg.flags : 0u; // no draw records at this level (shape cache): one state
	a.mdesc = (const VgxMeshDesc*)ctx->mdesc.p; a.draws = draws;
	a.mesh_cmd = a.jump0;
	a.partial = ctx->partial.p;
	a.max_meshes = meshCap;
	a.scan_bound = (out->meshes && out->cap_meshes && out->cap_meshes < meshCap) ? out->cap_meshes : meshCap; // (more meshes than the caller's table holds: VGX_E_NOSPACE was set by the scan over the meshes, the scan here covers nothing)
	a.uv = ctx->asmCfg.uv; a.uv_bytes = ctx->asmCfg.uv ? ctx->asmCfg.uv_bytes : 0u; a.uv_value[0] = ctx->asmCfg.uv_value[0]; a.uv_value[1] = ctx->asmCfg.uv_value[1];
	vgx_launch_assemble(a, s);
	mark(ctx, s, "assemble");
	return VGX_OK;
}

// tableDone: the scan over the meshes already wrote the caller's mesh table
int runStrokeEmit(vgx_ctx* ctx, const vgx_draw* draws, const vgx_mesh_out* out, hipStream_t s, const float* poly = nullptr, bool tableDone = false)
{
	if (ctx->asmArmed) {
		const int st = runAssemble(ctx, out, s, draws);
		if (st != VGX_OK) { return st; }
	}
	VgxStrokeArgs a;
	a.mesh_base = ctx->asmArmed ? (const uint32_t*)ctx->meshBase.p : nullptr;
	a.draws = draws; a.poly = poly ? poly : (const float*)ctx->poly.p; a.mdesc = (const VgxMeshDesc*)ctx->mdesc.p;
	a.elem_prefix = nullptr; a.elem_prefix_fill = (const uint64_t*)ctx->elemPrefix.p; a.elem_prefix_stroke = (const uint64_t*)ctx->elemPrefixS.p;
	a.mprep = (VgxMeshPrep*)ctx->mprep.p; a.mtab = (vgx_mesh*)ctx->mtab.p;
	a.pos = out->pos; a.color = out->color; a.idx = out->idx; a.meshes_out = tableDone ? nullptr : out->meshes;
	a.totals = (VgxTotals*)ctx->totals.p;
	a.caps = ctx->caps;
	a.tile_mode = 0;
	a.no_long = (out->cap_vertices < ctx->optBigEmitMin || !ctx->optStrokeLong) ? 1 : 0; // frame-sized: one stroke kernel less to launch
	// Batches of fills and closed Miter AA / Thin strokes (the scan over the meshes decides, on the device): one draw-ordered tile
	// kernel instead of k_fill + k_stroke_simple (vgx_tile.hip). Not for frame-sized calls (two more launches than they are worth).
	uint64_t capTiles = 0;
	if (ctx->optTileEmit && ctx->tileHint && !ctx->optConcurrentEmit && out->cap_vertices >= ctx->optBigEmitMin && out->cap_vertices / VGX_TILE_ELEMS + 2 < 0x7FFFFFFFull) {
		capTiles = out->cap_vertices / VGX_TILE_ELEMS + 2; // every element emits at least one vertex
		const int st = ensure(ctx, ctx->tileTab, capTiles * sizeof(VgxTileRec));
		if (st != VGX_OK) { return st; }
		a.tile_mode = 1;
	}
	if (ctx->optConcurrentEmit) {
		// tuning experiment (VGX_EXP_CONCURRENT_EMIT=1 at vgx_create): k_stroke on a side stream beside k_fill (they write disjoint
		// meshes); joined before the call returns control of `s`
		if (!ctx->sideStream) {
			(void)hipStreamCreateWithFlags(&ctx->sideStream, hipStreamNonBlocking);
			(void)hipEventCreateWithFlags(&ctx->forkEv, hipEventDisableTiming);
			(void)hipEventCreateWithFlags(&ctx->joinEv, hipEventDisableTiming);
		}
		(void)hipEventRecord(ctx->forkEv, s);
		(void)hipStreamWaitEvent(ctx->sideStream, ctx->forkEv, 0);
		VgxStrokeArgs b = a;
		b.elem_prefix = a.elem_prefix_stroke;
		vgx_launch_stroke(true, b, vgxElementGrid(out->cap_vertices), ctx->sideStream);
		(void)hipEventRecord(ctx->joinEv, ctx->sideStream);
		a.elem_prefix = a.elem_prefix_fill;
		vgx_launch_fill(a, vgxElementGrid(out->cap_vertices), s);
		(void)hipStreamWaitEvent(s, ctx->joinEv, 0);
		mark(ctx, s, "fill_emit");
		return launchStatus(ctx);
	}
	a.elem_prefix = a.elem_prefix_fill;
	vgx_launch_fill(a, vgxElementGrid(out->cap_vertices), s); // (tile mode: the mesh-table copy, and the kernel exits at once for the batches the tile kernel takes)
	mark(ctx, s, "fill_emit");
	a.elem_prefix = a.elem_prefix_stroke;
	vgx_launch_stroke(true, a, vgxElementGrid(out->cap_vertices), s);
	mark(ctx, s, "stroke_emit");
	if (a.tile_mode) {
		a.elem_prefix = nullptr;
		vgx_launch_emit_tiles(a, (VgxTileRec*)ctx->tileTab.p, capTiles, s);
		mark(ctx, s, "tile_emit");
	}
	return launchStatus(ctx);
}

int ensureMeshBuffers(vgx_ctx* ctx, uint64_t polyVerts, uint64_t subpaths, uint64_t meshes)
{
	int st;
	if ((st = ensure(ctx, ctx->poly, (polyVerts + 1) * 2 * sizeof(float))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->subs, (subpaths + 1) * sizeof(vgx_subpath))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->mdesc, (meshes + 1) * sizeof(VgxMeshDesc))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->elemPrefix, (meshes + 2) * sizeof(uint64_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->elemPrefixS, (meshes + 2) * sizeof(uint64_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->mprep, (meshes + 1) * sizeof(VgxMeshPrep))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->mtab, (meshes + 1) * sizeof(vgx_mesh))) != VGX_OK) { return st; }
	ctx->caps.poly_vertices = ctx->poly.cap / (2 * sizeof(float)) - 1;
	ctx->caps.subpaths = ctx->subs.cap / sizeof(vgx_subpath) - 1;
	uint64_t m = ctx->mdesc.cap / sizeof(VgxMeshDesc) - 1;
	uint64_t m2 = ctx->elemPrefix.cap / sizeof(uint64_t) - 2;
	{ const uint64_t m4 = ctx->elemPrefixS.cap / sizeof(uint64_t) - 2, m5 = ctx->mprep.cap / sizeof(VgxMeshPrep) - 1; if (m4 < m2) { m2 = m4; } if (m5 < m2) { m2 = m5; } }
	const uint64_t m3 = ctx->mtab.cap / sizeof(vgx_mesh) - 1;
	if (m2 < m) { m = m2; }
	if (m3 < m) { m = m3; }
	ctx->caps.meshes = m;
	return VGX_OK;
}

} // namespace

extern "C" {

uint32_t vgx_version(void) { return VGX_VERSION; }

const char* vgx_status_string(int status)
{
	switch (status) {
	case VGX_OK: return "VGX_OK";
	case VGX_E_INVALID_ARG: return "VGX_E_INVALID_ARG";
	case VGX_E_INVALID_PATH: return "VGX_E_INVALID_PATH";
	case VGX_E_NONFINITE: return "VGX_E_NONFINITE";
	case VGX_E_NOSPACE: return "VGX_E_NOSPACE";
	case VGX_E_MESH_TOO_LARGE: return "VGX_E_MESH_TOO_LARGE";
	case VGX_E_HIP: return "VGX_E_HIP";
	case VGX_E_NO_DEVICE: return "VGX_E_NO_DEVICE";
	case VGX_E_RANGE: return "VGX_E_RANGE";
	case VGX_E_INTERNAL: return "VGX_E_INTERNAL";
	case VGX_E_STALE: return "VGX_E_STALE";
	default: return "VGX_E_UNKNOWN";
	}
}

int vgx_create(int device, vgx_ctx** out_ctx)
{
	if (!out_ctx) {
		return VGX_E_INVALID_ARG;
	}
	*out_ctx = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
		return VGX_E_NO_DEVICE;
	}
	vgx_ctx* ctx = new (std::nothrow) vgx_ctx();
	if (!ctx) {
		return VGX_E_INVALID_ARG;
	}
	memset(ctx, 0, sizeof(*ctx));
	ctx->device = device;
	DeviceGuard guard(ctx); // the caller's current device is restored on return
	int cur = -1;
	if (hipGetDevice(&cur) != hipSuccess || cur != device) {
		delete ctx;
		return VGX_E_NO_DEVICE;
	}
	if (hipHostMalloc((void**)&ctx->hostTotals, sizeof(VgxTotals), hipHostMallocDefault) != hipSuccess) {
		delete ctx;
		return VGX_E_HIP;
	}
	// tuning / testing knobs: read once here, never on the call path
	ctx->optTwoPass = getenv("VGX_TWO_PASS_FLATTEN") ? 1 : 0;
	if (const char* e = getenv("VGX_PS_UPLOAD")) { ctx->optPsStage = strcmp(e, "stage") == 0; }
	ctx->optBigEmitMin = 1ull << 18;
	if (const char* e = getenv("VGX_BIG_EMIT_MIN")) { ctx->optBigEmitMin = strtoull(e, nullptr, 10); }
	ctx->optStrokeLong = 1;
	if (const char* e = getenv("VGX_STROKE_LONG")) { ctx->optStrokeLong = atoi(e) != 0; }
	ctx->optTileEmit = 1;
	if (const char* e = getenv("VGX_TILE_EMIT")) { ctx->optTileEmit = atoi(e) != 0; }
	ctx->optPsNoSmall = getenv("VGX_PS_NO_SMALL") ? 1 : 0; // testing knob: frame-sized path sets through the large-set launch sequence
	ctx->optBuildWaves = VGX_BUILD_WAVES;
	ctx->optConcurrentEmit = getenv("VGX_EXP_CONCURRENT_EMIT") ? 1 : 0;
	ctx->optNoSmall = getenv("VGX_NO_SMALL") ? 1 : 0; // testing knob: frame-sized batches through the large-batch launch sequence
	ctx->optInst = 1; ctx->optInstWaves = VGX_INST_WAVES; ctx->optInstBlock = VGX_INST_BLOCK; // VGX_INST=0: instanced batches through k_flatten_build as well
	if (const char* e = getenv("VGX_INST")) { ctx->optInst = atoi(e) != 0; }
	ctx->optInstClasses = 256; ctx->optInstPerm = 1;
	if (const char* e = getenv("VGX_INST_PERM")) { const int v = atoi(e); ctx->optInstPerm = v == 2 ? 2 : (v != 0); } // 2: the several-kernel sort for any count (testing)
	if (const char* e = getenv("VGX_INST_CLASSES")) { const int v = atoi(e); if (v >= 1 && v <= 65536) { uint32_t p2 = 1; while (p2 * 2 <= (uint32_t)v) { p2 *= 2; } ctx->optInstClasses = p2; } }
	if (const char* e = getenv("VGX_INST_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 65536) { ctx->optInstWaves = v; } }
	if (const char* e = getenv("VGX_INST_BLOCK")) { const int v = atoi(e); if (v >= 1 && v <= 65536) { ctx->optInstBlock = (uint32_t)v; } }
	ctx->optTmpl = 1; ctx->optTmplTile = VGX_TMPL_MAX_TILE; // VGX_TMPL=0: no template mode (instanced batches through k_flatten_inst + k_fill + k_stroke)
	if (const char* e = getenv("VGX_TMPL")) { ctx->optTmpl = atoi(e) != 0; }
	ctx->optTmplClasses = 1; // VGX_TMPL_CLASSES=0: template mode only for batches whose instances all repeat the first period
	if (const char* e = getenv("VGX_TMPL_CLASSES")) { ctx->optTmplClasses = atoi(e) != 0; }
	ctx->optTmplRound = 1;
	if (const char* e = getenv("VGX_TMPL_ROUND")) { ctx->optTmplRound = atoi(e) != 0; }
	ctx->optTmplBatch = 0;
	if (const char* e = getenv("VGX_TMPL_BATCH")) { ctx->optTmplBatch = atoi(e) != 0; }
	if (const char* e = getenv("VGX_TMPL_TILE")) { const int v = atoi(e); if (v >= 64 && v <= VGX_TMPL_MAX_TILE) { ctx->optTmplTile = (uint32_t)v / 64u * 64u; } } // testing: elements per tile (<= the LDS stage of k_tmpl_emit)
	ctx->optTessFlat1 = 1; ctx->f1Route = false; ctx->f1RoutePs = nullptr; ctx->f1RoutePsGen = 0; ctx->f1RouteCap = 1664; ctx->f1RouteSegMax = 64; ctx->f1RouteSegBound = 0;
	if (const char* e = getenv("VGX_TESS_FLAT1")) { const int v = atoi(e); if (v >= 0 && v <= 2) { ctx->optTessFlat1 = v; } }
	ctx->optPoolWalk = 0; // VGX_WALK=pool: the wave-cooperative walk of vgx_walk.h (same output, same speed: DESIGN.md section 4)
	if (const char* e = getenv("VGX_WALK")) { ctx->optPoolWalk = strcmp(e, "pool") == 0; }
	ctx->optF1Waves = 0; ctx->optF1Cap = 0; ctx->optF1Seg = 0;
	if (const char* e = getenv("VGX_F1_SEG")) { const int v = atoi(e); if (v >= 2 && v <= 64) { ctx->optF1Seg = v; } }
	if (const char* e = getenv("VGX_F1_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 65536) { ctx->optF1Waves = v; } }
	if (const char* e = getenv("VGX_F1_CAP")) { ctx->optF1Cap = atoi(e); }
	ctx->optThinStatic = 1;
	if (const char* e = getenv("VGX_THIN_STATIC")) { const int v = atoi(e); ctx->optThinStatic = v == 2 ? 2 : (v != 0 ? 1 : 0); } // (2: the kernel instance with two command instances per thread)
	if (const char* e = getenv("VGX_BUILD_WAVES")) { const int v = atoi(e); if (v >= 1 && v < VGX_BUILD_WAVES) { ctx->optBuildWaves = v; } }
	*out_ctx = ctx;
	return VGX_OK;
}

int vgx_destroy(vgx_ctx* ctx)
{
	if (!ctx) {
		return VGX_E_INVALID_ARG;
	}
	DeviceGuard guard(ctx);
	DevBuf* bufs[] = { &ctx->tmplClsSum, &ctx->tileTab, &ctx->psTemp, &ctx->f1SegDraw, &ctx->f1Segs, &ctx->tmplHash, &ctx->tmplInstCls, &ctx->tmplClsRep, &ctx->tmplCls, &ctx->tmplIinfo, &ctx->tmplWg, &ctx->tmplTrmesh, &ctx->tmplTmsz, &ctx->tmplRsz, &ctx->tmplRelem, &ctx->tmplMplace, &ctx->tmplItot, &ctx->tmplIplace, &ctx->tmplTile, &ctx->tmplPoly, &ctx->tmplMesh, &ctx->tmplMtab, &ctx->tmplElem, &ctx->tmplDraws, &ctx->partBounds, &ctx->instPerm, &ctx->instPermHist, &ctx->instHist, &ctx->instCursor, &ctx->instKeyStart, &ctx->instStart, &ctx->instTaskStart, &ctx->instTaskPath, &ctx->instOrder, &ctx->gatherSizes, &ctx->asmJump0, &ctx->asmJump1, &ctx->asmStart, &ctx->meshBase, &ctx->subPrefix, &ctx->cmdPrefix, &ctx->cmdCnt, &ctx->subFirst, &ctx->leafOverflow, &ctx->serialList, &ctx->dinfo, &ctx->poly, &ctx->subs, &ctx->mdesc, &ctx->elemPrefix, &ctx->elemPrefixS, &ctx->mprep, &ctx->mtab, &ctx->partial, &ctx->totals };
	for (DevBuf* b : bufs) {
		if (b->p) { (void)hipFree(b->p); }
	}
	if (ctx->hostTotals) { (void)hipHostFree(ctx->hostTotals); }
	if (ctx->hostF1) { (void)hipHostFree(ctx->hostF1); }
	if (ctx->hostPs) { (void)hipHostFree(ctx->hostPs); }
	if (ctx->psImage) { (void)hipHostFree(ctx->psImage); }
	for (int i = 0; i < VGX_PS_POOL; ++i) { if (ctx->psPool[i].p) { (void)hipFree(ctx->psPool[i].p); } }
	if (ctx->psStream) { (void)hipStreamDestroy(ctx->psStream); }
	for (int i = 0; i < 2; ++i) { if (ctx->psStage[i]) { (void)hipHostFree(ctx->psStage[i]); (void)hipEventDestroy(ctx->psStageEv[i]); } }
	vgx_rccl_release(ctx);
	if (ctx->sideStream) { (void)hipStreamDestroy(ctx->sideStream); (void)hipEventDestroy(ctx->forkEv); (void)hipEventDestroy(ctx->joinEv); }
	if (ctx->evCreated) {
		for (int r = 0; r < VGX_PROF_RING; ++r) { for (int i = 0; i <= VGX_MAX_STAGES; ++i) { (void)hipEventDestroy(ctx->ev[r][i]); } }
	}
	delete ctx;
	return VGX_OK;
}

int vgx_last_hip_error(const vgx_ctx* ctx) { return ctx ? ctx->lastHipError : 0; }

uint64_t vgx_scratch_bytes(const vgx_ctx* ctx)
{
	if (!ctx) {
		return 0;
	}
	return ctx->tmplClsSum.cap + ctx->tileTab.cap + ctx->psTemp.cap + ctx->f1SegDraw.cap + ctx->f1Segs.cap + ctx->tmplHash.cap + ctx->tmplInstCls.cap + ctx->tmplClsRep.cap + ctx->tmplCls.cap + ctx->tmplIinfo.cap + ctx->tmplWg.cap + ctx->tmplTrmesh.cap + ctx->tmplTmsz.cap + ctx->tmplRsz.cap + ctx->tmplRelem.cap + ctx->tmplMplace.cap + ctx->tmplItot.cap + ctx->tmplIplace.cap + ctx->tmplTile.cap + ctx->tmplPoly.cap + ctx->tmplMesh.cap + ctx->tmplMtab.cap + ctx->tmplElem.cap + ctx->tmplDraws.cap + ctx->gatherSizes.cap + ctx->asmJump0.cap + ctx->asmJump1.cap + ctx->asmStart.cap + ctx->meshBase.cap + ctx->subPrefix.cap + ctx->cmdPrefix.cap + ctx->cmdCnt.cap + ctx->subFirst.cap + ctx->leafOverflow.cap + ctx->serialList.cap + ctx->dinfo.cap + ctx->poly.cap + ctx->subs.cap + ctx->mdesc.cap + ctx->elemPrefix.cap + ctx->elemPrefixS.cap + ctx->mprep.cap + ctx->mtab.cap + ctx->partial.cap + ctx->totals.cap;
}

// ---- path set ---------------------------------------------------------------------------------------
// Host-side validation of the command grammar (see include/vgx.h) and derivation of the static
// per-command structure the kernels use (sub-path heads / tails, serial-path flag).
} // extern "C"

#include "vgx_pathset_host.h"

extern "C" {

int vgx_pathset_validate(const vgx_pathset_desc* desc)
{
	std::vector<uint8_t> cmdFlags, pathFlags;
	std::vector<uint32_t> spStart;
	uint32_t maxCmds = 0;
	return vgx_pathset_validate_host(desc, &cmdFlags, &spStart, &pathFlags, &maxCmds);
}

// The caller's arrays are ordinary host memory: hipMemcpyAsync moves them (the runtime pins pageable ranges in place for large
// copies: 22 GB/s cold, 55 GB/s for a range it has seen, profiles/micro/h2d_probe.hip), or -- VGX_PS_UPLOAD=stage at vgx_create --
// two 8 MB pinned buffers of the context, filled by memcpy while the other one is on the wire (28 GB/s whatever the runtime does).
static int psUpload(vgx_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s)
{
	if (!bytes) { return VGX_OK; }
	if (!ctx->optPsStage) {
		HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
		return VGX_OK;
	}
	const size_t chunk = (size_t)8 << 20;
	for (int i = 0; i < 2; ++i) {
		if (!ctx->psStage[i]) {
			HIPCHK(ctx, hipHostMalloc(&ctx->psStage[i], chunk, hipHostMallocDefault));
			HIPCHK(ctx, hipEventCreateWithFlags(&ctx->psStageEv[i], hipEventDisableTiming));
			ctx->psStageBusy[i] = false;
		}
	}
	for (size_t o = 0; o < bytes; o += chunk) {
		const int i = (int)(ctx->psStageK++ & 1u);
		const size_t n = bytes - o < chunk ? bytes - o : chunk;
		if (ctx->psStageBusy[i]) { HIPCHK(ctx, hipEventSynchronize(ctx->psStageEv[i])); }
		memcpy(ctx->psStage[i], (const uint8_t*)src + o, n);
		HIPCHK(ctx, hipMemcpyAsync((uint8_t*)dst + o, ctx->psStage[i], n, hipMemcpyHostToDevice, s));
		HIPCHK(ctx, hipEventRecord(ctx->psStageEv[i], s));
		ctx->psStageBusy[i] = true;
	}
	return VGX_OK;
}

struct VgxPsLayout
{
	size_t oArgs, oArgOff, oPathBegin, oType, rawEnd, oSpStart, oFlags, oPathFlags, oRec, oSubBegin, oSubLast, oThin, oThinPath, oThinSub, total;
};
// One blob: the four raw arrays first, back to back ([args (2 floats of padding in front: the start-point gather of command 0
// reads args[-2..-1])] [arg offsets] [path begins] [opcodes] -- a frame-sized set goes up as ONE image of this part), the derived
// tables behind them. The sub-path tables are sized for the most sub-paths `ncmd` commands can hold (every command ends one): what
// the set really has is known only after the scan, and 12 bytes per command of head room are nothing against 288 GB.
static VgxPsLayout psLayout(uint32_t ncmd, uint32_t npaths, uint32_t nargs)
{
	auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
	VgxPsLayout L;
	L.oArgs = 0;
	L.oArgOff = align(L.oArgs + ((size_t)nargs + 4) * sizeof(float));
	L.oPathBegin = align(L.oArgOff + ((size_t)ncmd + 1) * sizeof(uint32_t));
	L.oType = align(L.oPathBegin + ((size_t)npaths + 1) * sizeof(uint32_t));
	L.rawEnd = align(L.oType + ncmd + 1);
	L.oSpStart = L.rawEnd;
	L.oFlags = align(L.oSpStart + ((size_t)ncmd + 1) * sizeof(uint32_t));
	L.oPathFlags = align(L.oFlags + ncmd + 1);
	L.oRec = align(L.oPathFlags + npaths + 4);
	L.oSubBegin = align(L.oRec + ((size_t)ncmd + 1) * sizeof(VgxCmdRec));
	L.oSubLast = align(L.oSubBegin + ((size_t)npaths + 1) * sizeof(uint32_t));
	L.oThin = align(L.oSubLast + ((size_t)ncmd + 1) * sizeof(uint32_t));
	L.oThinPath = align(L.oThin + ((size_t)ncmd + 3) * sizeof(VgxCmdThin));
	L.oThinSub = align(L.oThinPath + ((size_t)npaths + 1) * sizeof(VgxThinPath));
	L.total = align(L.oThinSub + ((size_t)ncmd + 1) * sizeof(VgxThinSub));
	return L;
}

// Blobs of frame-sized sets are recycled through the context (hipFree synchronises the device; a renderer makes and drops one
// such set per frame): up to VGX_PS_POOL blobs of at most VGX_PS_POOL_MAX bytes wait here for the next vgx_pathset_create.
static void* psPoolTake(vgx_ctx* ctx, size_t need, size_t* got)
{
	int best = -1;
	for (int i = 0; i < VGX_PS_POOL; ++i) {
		if (ctx->psPool[i].p && ctx->psPool[i].cap >= need && (best < 0 || ctx->psPool[i].cap < ctx->psPool[best].cap)) { best = i; }
	}
	if (best < 0) { return nullptr; }
	void* p = ctx->psPool[best].p;
	*got = ctx->psPool[best].cap;
	ctx->psPool[best].p = nullptr; ctx->psPool[best].cap = 0;
	return p;
}
static bool psPoolGive(vgx_ctx* ctx, void* p, size_t cap)
{
	if (cap > VGX_PS_POOL_MAX) { return false; }
	for (int i = 0; i < VGX_PS_POOL; ++i) {
		if (!ctx->psPool[i].p) { ctx->psPool[i].p = p; ctx->psPool[i].cap = cap; return true; }
	}
	return false;
}

// Round 6: the raw arrays go up as they are and the derived tables are built by kernels (vgx_pathset.hip): grammar checks as a
// flagged reduction, sub-path heads / path of a command / sub-path ordinals by ONE scan over the commands, the 64-byte command
// records and the thin records by a lane per command, the static polyline layout of lineTo-only sets by a second scan. The host
// reads 32 bytes at the end. An invalid set takes the slow path: the host validator names the status.
int vgx_pathset_create(vgx_ctx* ctx, const vgx_pathset_desc* desc, vgx_pathset** out_ps)
{
	DeviceGuard guard(ctx);
	if (!ctx || !desc || !out_ps) {
		return VGX_E_INVALID_ARG;
	}
	*out_ps = nullptr;
	// the O(1) checks of the validator's head (vgx_pathset_host.h): everything the layout below depends on
	if (!desc->path_cmd_begin || !desc->cmd_arg_off || (desc->ncmd && !desc->cmd_type)) { return VGX_E_INVALID_ARG; }
	if (desc->path_cmd_begin[0] != 0 || desc->path_cmd_begin[desc->npaths] != desc->ncmd || desc->cmd_arg_off[0] != 0) { return VGX_E_INVALID_ARG; }
	if (desc->ncmd >= 0x7FFFFFFFu) { return VGX_E_INVALID_ARG; } // bit 31 of a command index carries a flag in the draw window
	const uint32_t ncmd = desc->ncmd, npaths = desc->npaths;
	const uint32_t nargs = desc->cmd_arg_off[ncmd];
	if (nargs && !desc->args) { return VGX_E_INVALID_ARG; }
	const VgxPsLayout L = psLayout(ncmd, npaths, nargs);
	int st;
	static const bool timing = getenv("VGX_PS_TIMING") != nullptr; // tuning aid: host clock after each stage of the call, to stderr
	double tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
	if (timing) { tq[0] = now(); }
	// temporaries of the build (grow-only context scratch): four words per command, the scans' partials, the totals
	auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
	// layout: [totals (256 B) | pathAt] (cleared by ONE memset) | pathOf | lastSubEx | nvEx | partials
	const size_t words = a256(((size_t)ncmd + 1) * sizeof(uint32_t));
	const size_t tTot = 0, tW = 256, tPartA = tW + 4 * words, tPartB = tPartA + a256(VGX_MSCAN_BLOCKS * sizeof(VgxPsM)), tEnd = tPartB + a256(VGX_SCAN_BLOCKS * sizeof(Sum3));
	if ((st = ensure(ctx, ctx->psTemp, tEnd)) != VGX_OK) { return st; }
	if (!ctx->psStream) { HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->psStream, hipStreamNonBlocking)); }
	if (!ctx->hostPs) { HIPCHK(ctx, hipHostMalloc((void**)&ctx->hostPs, sizeof(VgxPsTotals), hipHostMallocDefault)); }
	hipStream_t s = ctx->psStream;

	vgx_pathset* ps = new (std::nothrow) vgx_pathset();
	if (!ps) {
		return VGX_E_INVALID_ARG;
	}
	const bool small = !ctx->optPsNoSmall && vgx_pathset_is_small(ncmd, npaths, nargs);
	ps->blob = nullptr;
	ps->blobBytes = L.total;
	if (small) { ps->blob = psPoolTake(ctx, L.total, &ps->blobBytes); }
	hipError_t e = ps->blob ? hipSuccess : hipMalloc(&ps->blob, L.total);
	if (e != hipSuccess) {
		ctx->lastHipError = (int)e;
		delete ps;
		return VGX_E_HIP;
	}
	if (timing) { tq[1] = now(); }
	uint8_t* b = (uint8_t*)ps->blob;
	uint8_t* t = (uint8_t*)ctx->psTemp.p;
	auto fail = [&](int code) { (void)hipStreamSynchronize(s); (void)hipFree(ps->blob); delete ps; return code; };
	// totals and path marks start from zero (one memset: they are neighbours); what the passes OR into inside the blob and the padding
	// records the flatten kernels read past the ends are cleared by the first kernel -- no memset of the blob (1 GB for configs[3])
	e = hipMemsetAsync(t + tTot, 0, tW + words, s);
	if (e != hipSuccess) { ctx->lastHipError = (int)e; return fail(VGX_E_HIP); }
	bool maybeThin = true;
	if (small) {
		for (uint32_t c = 0; c < ncmd && maybeThin; ++c) { const uint8_t ty = desc->cmd_type[c]; maybeThin = ty == VGX_CMD_MOVE_TO || ty == VGX_CMD_LINE_TO || ty == VGX_CMD_CLOSE; }
		// ONE image of the raw part, filled in pinned memory, one copy
		if (L.rawEnd > ctx->psImageCap) {
			if (ctx->psImage) { (void)hipHostFree(ctx->psImage); ctx->psImage = nullptr; ctx->psImageCap = 0; }
			const size_t want = L.rawEnd * 2 > ((size_t)1 << 16) ? L.rawEnd * 2 : ((size_t)1 << 16);
			e = hipHostMalloc(&ctx->psImage, want, hipHostMallocDefault);
			if (e != hipSuccess) { ctx->lastHipError = (int)e; return fail(VGX_E_HIP); }
			ctx->psImageCap = want;
		}
		uint8_t* im = (uint8_t*)ctx->psImage;
		memset(im + L.oArgs, 0, 2 * sizeof(float));
		if (nargs) { memcpy(im + L.oArgs + 2 * sizeof(float), desc->args, (size_t)nargs * sizeof(float)); }
		memcpy(im + L.oArgOff, desc->cmd_arg_off, ((size_t)ncmd + 1) * sizeof(uint32_t));
		memcpy(im + L.oPathBegin, desc->path_cmd_begin, ((size_t)npaths + 1) * sizeof(uint32_t));
		if (ncmd) { memcpy(im + L.oType, desc->cmd_type, ncmd); }
		im[L.oType + ncmd] = 0;
		if (timing) { tq[2] = now(); }
		e = hipMemcpyAsync(b, im, L.rawEnd, hipMemcpyHostToDevice, s);
		if (e != hipSuccess) { ctx->lastHipError = (int)e; return fail(VGX_E_HIP); }
	} else {
		if ((st = psUpload(ctx, b + L.oArgs + 2 * sizeof(float), desc->args, (size_t)nargs * sizeof(float), s)) != VGX_OK) { return fail(st); }
		if ((st = psUpload(ctx, b + L.oArgOff, desc->cmd_arg_off, ((size_t)ncmd + 1) * sizeof(uint32_t), s)) != VGX_OK) { return fail(st); }
		if ((st = psUpload(ctx, b + L.oType, desc->cmd_type, ncmd, s)) != VGX_OK) { return fail(st); }
		if ((st = psUpload(ctx, b + L.oPathBegin, desc->path_cmd_begin, ((size_t)npaths + 1) * sizeof(uint32_t), s)) != VGX_OK) { return fail(st); }
	}
	VgxPsBuild B;
	B.type = b + L.oType; B.argOff = (const uint32_t*)(b + L.oArgOff); B.args = (const float*)(b + L.oArgs) + 2; B.pcb = (const uint32_t*)(b + L.oPathBegin);
	B.ncmd = ncmd; B.npaths = npaths; B.nargs = nargs;
	B.spStart = (uint32_t*)(b + L.oSpStart); B.flags = b + L.oFlags; B.pathFlags = b + L.oPathFlags; B.rec = (VgxCmdRec*)(b + L.oRec);
	B.thin = (VgxCmdThin*)(b + L.oThin) + 1; // thin[-1]: padding record
	B.subBegin = (uint32_t*)(b + L.oSubBegin); B.subLast = (uint32_t*)(b + L.oSubLast); B.tp = (VgxThinPath*)(b + L.oThinPath); B.ts = (VgxThinSub*)(b + L.oThinSub);
	B.pathAt = (uint32_t*)(t + tW); B.pathOf = (uint32_t*)(t + tW + words); B.lastSubEx = (uint32_t*)(t + tW + 2 * words); B.nvEx = (uint32_t*)(t + tW + 3 * words);
	B.partialA = (VgxPsM*)(t + tPartA); B.partialB = (Sum3*)(t + tPartB); B.tot = (VgxPsTotals*)(t + tTot);
	if (timing) { tq[3] = now(); }
	vgx_launch_pathset_build(B, maybeThin, s);
	e = hipGetLastError();
	if (timing) { tq[4] = now(); }
	if (e == hipSuccess) { e = hipMemcpyAsync(ctx->hostPs, B.tot, sizeof(VgxPsTotals), hipMemcpyDeviceToHost, s); }
	if (timing) { tq[5] = now(); }
	if (e == hipSuccess) { e = hipStreamSynchronize(s); }
	if (timing) {
		tq[6] = now();
		fprintf(stderr, "vgx_pathset_create ncmd=%u small=%d: alloc %.1f  image %.1f  h2d %.1f  launch %.1f  d2h %.1f  sync %.1f  total %.1f us\n", ncmd, (int)small,
			tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4], tq[6] - tq[5], tq[6] - tq[0]);
	}
	if (e != hipSuccess) { ctx->lastHipError = (int)e; return fail(VGX_E_HIP); }
	const VgxPsTotals T = *ctx->hostPs;
	if (T.err) {
		(void)hipFree(ps->blob);
		delete ps;
		const int vst = vgx_pathset_validate(desc); // the slow path of an invalid set: which status it is
		return vst != VGX_OK ? vst : VGX_E_INTERNAL;
	}
	ps->maxCmdsPerPath = T.maxCmds;
	ps->hasSerial = T.hasSerial != 0;
	ps->hasEmpty = T.hasEmpty != 0;
	ps->numSubs = T.nsubs;
	ps->thinStatic = npaths != 0 && ncmd != 0 && !T.notThin && !T.hasSerial && !T.hasEmpty && !T.thinIneligible;
	ps->dev.args = (const float*)(b + L.oArgs) + 2;
	ps->dev.cmd_arg_off = (const uint32_t*)(b + L.oArgOff);
	ps->dev.cmd_sp_start = (const uint32_t*)(b + L.oSpStart);
	ps->dev.path_cmd_begin = (const uint32_t*)(b + L.oPathBegin);
	ps->dev.cmd_type = b + L.oType;
	ps->dev.cmd_flags = b + L.oFlags;
	ps->dev.path_flags = b + L.oPathFlags;
	ps->dev.cmdrec = (const VgxCmdRec*)(b + L.oRec);
	ps->dev.cmdthin = (const VgxCmdThin*)(b + L.oThin) + 1;
	ps->dev.thin_path = (const VgxThinPath*)(b + L.oThinPath);
	ps->dev.thin_sub = (const VgxThinSub*)(b + L.oThinSub);
	ps->dev.path_sub_begin = (const uint32_t*)(b + L.oSubBegin);
	ps->dev.sub_last_cmd = (const uint32_t*)(b + L.oSubLast);
	ps->dev.npaths = npaths;
	ps->dev.ncmd = ncmd;
	static std::atomic<uint64_t> s_pathsetGen{0};
	ps->gen = ++s_pathsetGen;
	*out_ps = ps;
	return VGX_OK;
}

// Testing / inspection: copies one of the set's device tables to host memory (tests/test_gpu_pathset_build.py compares every table
// with the host restatement of csrc/vgx_pathset_host.h, byte for byte). `bytes` receives the table's size; dst may be NULL to ask.
int vgx_pathset_read_table(vgx_ctx* ctx, const vgx_pathset* ps, int which, void* dst, uint64_t cap_bytes, uint64_t* bytes)
{
	DeviceGuard guard(ctx);
	if (!ctx || !ps || !bytes) { return VGX_E_INVALID_ARG; }
	const VgxPathSetDev& d = ps->dev;
	const void* src = nullptr; uint64_t n = 0;
	uint32_t scal[8] = { ps->maxCmdsPerPath, ps->hasSerial ? 1u : 0u, ps->hasEmpty ? 1u : 0u, ps->thinStatic ? 1u : 0u, ps->numSubs, d.npaths, d.ncmd, 0u };
	switch (which) {
	case VGX_PS_TABLE_CMD_FLAGS: src = d.cmd_flags; n = d.ncmd; break;
	case VGX_PS_TABLE_SP_START: src = d.cmd_sp_start; n = (uint64_t)d.ncmd * 4; break;
	case VGX_PS_TABLE_PATH_FLAGS: src = d.path_flags; n = d.npaths; break;
	case VGX_PS_TABLE_CMDREC: src = d.cmdrec; n = (uint64_t)d.ncmd * sizeof(VgxCmdRec); break;
	case VGX_PS_TABLE_PATH_SUB_BEGIN: src = d.path_sub_begin; n = ((uint64_t)d.npaths + 1) * 4; break;
	case VGX_PS_TABLE_SUB_LAST_CMD: src = d.sub_last_cmd; n = (uint64_t)ps->numSubs * 4; break;
	case VGX_PS_TABLE_CMDTHIN: src = d.cmdthin; n = (uint64_t)d.ncmd * sizeof(VgxCmdThin); break;
	case VGX_PS_TABLE_THIN_PATH: src = d.thin_path; n = ps->thinStatic ? (uint64_t)d.npaths * sizeof(VgxThinPath) : 0; break;
	case VGX_PS_TABLE_THIN_SUB: src = d.thin_sub; n = ps->thinStatic ? (uint64_t)ps->numSubs * sizeof(VgxThinSub) : 0; break;
	case VGX_PS_TABLE_SCALARS: n = sizeof(scal); break;
	default: return VGX_E_INVALID_ARG;
	}
	*bytes = n;
	if (!dst) { return VGX_OK; }
	if (cap_bytes < n) { return VGX_E_NOSPACE; }
	if (which == VGX_PS_TABLE_SCALARS) { memcpy(dst, scal, sizeof(scal)); return VGX_OK; }
	if (n) { HIPCHK(ctx, hipMemcpy(dst, src, n, hipMemcpyDeviceToHost)); }
	return VGX_OK;
}

int vgx_pathset_destroy(vgx_ctx* ctx, vgx_pathset* ps)
{
	DeviceGuard guard(ctx);
	if (!ctx || !ps) {
		return VGX_E_INVALID_ARG;
	}
	if (ctx->lastPs == ps) { ctx->lastPs = nullptr; ctx->lastStage = 0; }
	if (ctx->tmplPs == ps) { ctx->tmplOn = false; ctx->tmplPs = nullptr; ctx->tmplPsGen = 0; } // a later path set at the same address must not match the template
	// hipFree waits for everything in flight (a caller may drop a set right behind an asynchronous call that reads it); a blob that
	// goes back to the pool instead must be as idle: the same wait, without the allocator's work
	if (ps->blob && ps->blobBytes <= VGX_PS_POOL_MAX) { (void)hipDeviceSynchronize(); }
	if (ps->blob && !psPoolGive(ctx, ps->blob, ps->blobBytes)) { (void)hipFree(ps->blob); }
	delete ps;
	return VGX_OK;
}

// ---- flatten ------------------------------------------------------------------------------------------
// detectInst: look for reused paths (instanced flatten kernel of vgx_tessellate); the flatten-only entry points have no use for it
static int flattenCountCommon(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, hipStream_t s, bool detectInst)
{
	int st;
	if ((st = ensureDrawBuffers(ctx, ndraws)) != VGX_OK) { return st; }
	// pass 1: command instances (sizes the per-command scratch)
	ctx->caps.cmd_instances = ~0ull; // not known yet: never trips the check in this sizing pass
	ctx->instPeriod = 0; ctx->instGrouped = 0; ctx->instClasses = 1; ctx->instPermOn = 0;
	ctx->tmplOn = false; // the scratch is re-sized for this batch; a template belongs to the count call that built it
	runCmdPrefix(ctx, ps, draws, ndraws, s);
	if (detectInst && ctx->optInst && ndraws > VGX_SMALL_DRAWS) {
		vgx_launch_inst_detect(draws, ndraws, (VgxTotals*)ctx->totals.p, s);
		// ... and how many different paths the draws use (grouped mode, when the sequence does not repeat)
		if ((st = ensureInstGroup(ctx, ps->dev.npaths, 1, ndraws, false)) != VGX_OK) { return st; }
		vgx_launch_inst_group(draws, ndraws, ps->dev.npaths, 1, (uint32_t*)ctx->instHist.p, nullptr, nullptr, (uint64_t*)ctx->instStart.p, (uint64_t*)ctx->instTaskStart.p,
			nullptr, 0, nullptr, (VgxTotals*)ctx->totals.p, ctx->partial.p, s);
	}
	if ((st = readTotals(ctx, s)) != VGX_OK) { return st; }
	if (ctx->hostTotals->status != VGX_OK) { return (int)ctx->hostTotals->status; }
	if (ctx->hostTotals->inst_detect_inv != 0 && !ctx->hostTotals->inst_detect_bad) {
		// the draws repeat one sequence of paths (a drawing submitted for many instances): vgx_inst.hip
		const unsigned long long P = ~0ull - ctx->hostTotals->inst_detect_inv;
		if (P <= 0xFFFFFFFFull) { ctx->instPeriod = (uint32_t)P; }
	}
	const VgxTotals& ht = *ctx->hostTotals;
	const bool reused = ctx->optInst && ndraws > VGX_SMALL_DRAWS && ndraws < 0xFFFFFFFFull && ht.inst_distinct != 0 && ndraws / ht.inst_distinct >= VGX_INST_MIN_INSTANCES;
	if (reused && ctx->optInstClasses > 1 && (instPeriodFor(ctx, ndraws) ? ht.inst_tol_varies != 0 : (uint32_t)~ht.inst_tol_lo_inv != ht.inst_tol_hi)) {
		// The instances of a path differ in scale: in the periodic mapping (lane = instance) the lanes of a wave would
		// disagree about nearly every cubic. Grouped mode sorted by (path, tolerance class) instead.
		uint32_t nc = ctx->optInstClasses;
		if (instPeriodFor(ctx, ndraws) && ctx->optInstPerm) {
			// the sequence repeats: keep the periodic mapping (closed-form offsets, no per-draw sort) and permute the INSTANCES
			const uint64_t ninst = ndraws / ctx->instPeriod;
			if (nc > 1024u) { nc = 1024u; }
			if ((st = ensure(ctx, ctx->instPerm, (ninst + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
			if ((st = ensure(ctx, ctx->instPermHist, ((size_t)nc + 2) * sizeof(uint32_t))) != VGX_OK) { return st; }
			ctx->instClasses = nc;
			ctx->instPermOn = 1;
		} else {
			while (nc > 1 && (uint64_t)ps->dev.npaths * nc > (1ull << 22)) { nc >>= 1; } // scratch of at most 4 M keys
			ctx->instClasses = nc;
			if (nc > 1) { ctx->instPeriod = 0; }
		}
	}
	if (!instPeriodFor(ctx, ndraws) && ctx->optInst && ndraws > VGX_SMALL_DRAWS && ndraws < 0xFFFFFFFFull && ctx->hostTotals->inst_distinct != 0
		&& ndraws / ctx->hostTotals->inst_distinct >= VGX_INST_MIN_INSTANCES) {
		// paths are reused (>= 32 draws per used path on average) but not as a repeating sequence: grouped mode
		if ((st = ensureInstGroup(ctx, ps->dev.npaths, ctx->instClasses, ndraws, true)) != VGX_OK) { return st; }
		ctx->instGrouped = 1;
	}
	const uint64_t ncmdInst = ctx->hostTotals->sizes.num_cmd_instances;
	if ((st = ensure(ctx, ctx->cmdCnt, (ncmdInst + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->subFirst, (ncmdInst + 1) * sizeof(VgxSubRec))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->leafOverflow, (size_t)VGX_BUILD_WAVES * VGX_BUILD_OVERFLOW * 64 * 2 * sizeof(float))) != VGX_OK) { return st; }
	ctx->caps.cmd_instances = ctx->cmdCnt.cap / sizeof(uint32_t) - 1;
	{ const uint64_t c2 = ctx->subFirst.cap / sizeof(VgxSubRec) - 1; if (c2 < ctx->caps.cmd_instances) { ctx->caps.cmd_instances = c2; } }
	// pass 2: per-draw counts (sizes polyline / sub-path / mesh scratch)
	const VgxCaps saved = ctx->caps;
	ctx->caps.poly_vertices = ~0ull; ctx->caps.subpaths = ~0ull; ctx->caps.meshes = ~0ull;
	runFlattenCount(ctx, ps, draws, ndraws, s);
	ctx->caps = saved;
	if ((st = readTotals(ctx, s)) != VGX_OK) { return st; }
	if (ctx->hostTotals->status != VGX_OK) { return (int)ctx->hostTotals->status; }
	if (ctx->hostTotals->sizes.num_poly_vertices > 0xFFFFFFF0ull) { return VGX_E_RANGE; }
	return VGX_OK;
}

int vgx_flatten_count(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, vgx_sizes* out_sizes, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !ps || (!draws && ndraws) || !out_sizes) {
		return VGX_E_INVALID_ARG;
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	ctx->lastStage = 0;
	const int st = flattenCountCommon(ctx, ps, draws, ndraws, s, false);
	if (st != VGX_OK) { return st; }
	*out_sizes = ctx->hostTotals->sizes;
	out_sizes->num_vertices = 0;
	out_sizes->num_indices = 0;
	ctx->lastPs = ps; ctx->lastDraws = draws; ctx->lastNDraws = ndraws; ctx->lastStage = 1;
	return VGX_OK;
}

int vgx_flatten_emit(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, int apply_transform, const vgx_flat_out* out, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !ps || !out || (!draws && ndraws)) {
		return VGX_E_INVALID_ARG;
	}
	if (ctx->lastStage != 1 || ctx->lastPs != ps || ctx->lastDraws != draws || ctx->lastNDraws != ndraws) {
		return VGX_E_INVALID_ARG; // must follow vgx_flatten_count on the same batch (not vgx_tessellate_count: its polyline scratch is live)
	}
	const vgx_sizes& sz = ctx->hostTotals->sizes;
	if ((out->poly && out->cap_poly_vertices < sz.num_poly_vertices) || (out->subpaths && out->cap_subpaths < sz.num_subpaths)) {
		return VGX_E_NOSPACE;
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	int st;
	// outputs the caller does not want still need somewhere to go
	if (!out->poly) { if ((st = ensure(ctx, ctx->poly, (sz.num_poly_vertices + 1) * 2 * sizeof(float))) != VGX_OK) { return st; } }
	if (!out->subpaths) { if ((st = ensure(ctx, ctx->subs, (sz.num_subpaths + 1) * sizeof(vgx_subpath))) != VGX_OK) { return st; } }
	VgxFlattenArgs a = flattenArgs(ctx, ps, draws, ndraws, apply_transform);
	a.poly = out->poly ? out->poly : (float*)ctx->poly.p;
	a.subs = out->subpaths ? out->subpaths : (vgx_subpath*)ctx->subs.p;
	a.mdesc = nullptr;
	a.mtab = nullptr;
	a.caps.poly_vertices = ~0ull; a.caps.subpaths = ~0ull; a.caps.meshes = ~0ull;
	vgx_launch_flatten(true, a, VGX_GRID_BLOCKS, s);
	mark(ctx, s, "flatten_emit");
	if (out->draw_info && ndraws) {
		HIPCHK(ctx, hipMemcpyAsync(out->draw_info, ctx->dinfo.p, ndraws * sizeof(vgx_draw_info), hipMemcpyDeviceToDevice, s));
	}
	return VGX_OK;
}

// ---- vgx_flatten: the ordered one-walk flatten (vgx_flat1.hip), asynchronous ---------------------------------------------
int vgx_flatten(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, int apply_transform, const vgx_flat_out* out,
                vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !ps || !out || !out->poly || !out->subpaths || (!draws && ndraws)) {
		return VGX_E_INVALID_ARG;
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	ctx->lastStage = 0;
	ctx->tmplOn = false;
	int st;
	if ((st = ensureDrawBuffers(ctx, ndraws)) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->serialList, (ndraws + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
	// Leaf-list capacity of the kernel instance and commands per segment bucket. A 64-command chunk of ~45-segment cubics stages
	// ~1450 leaves (the 1664-entry instance: six waves per CU); lighter batches take the 1024-entry instance (eight waves per CU),
	// heavier ones (long curves: 145 segments per cubic) the 3072-entry instance with smaller buckets, so that a chunk's leaves
	// still fit (a chunk that overflows the list is placed by walking its cubics a second time). The batch's leaves per command are
	// taken from the LAST call on this context with the same path set and draw count, when that call's result has arrived.
	int cap = 1664; uint32_t segMax = 64;
	const uint64_t tag = ps->gen * 0x9E3779B97F4A7C15ull + ndraws;
	if (!ctx->hostF1) {
		if (hipHostMalloc((void**)&ctx->hostF1, 4 * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess) { ctx->hostF1[0] = 0; ctx->hostF1[1] = 0; ctx->hostF1[2] = 0; ctx->hostF1[3] = 0; }
		else { ctx->hostF1 = nullptr; }
	}
	if (ctx->hostF1 && ctx->f1Tag == tag && ctx->hostF1[1] != 0 && ctx->hostF1[2] == tag) {
		const double perCmd = (double)ctx->hostF1[0] / (double)ctx->hostF1[1];
		if (perCmd * 64.0 <= 800.0) { cap = 1024; }
		else if (perCmd * 64.0 > 1500.0) {
			// long curves (145 segments per cubic at box 10 000): the 3072-entry instance with buckets of a power of two -- 2^k cubics
			// become 64 tasks after whole rounds of cutting (17 cubics would stay at 34 half-cubic tasks). Measured on 1 M such
			// cubics: 3072 entries x 32 commands 3.4 ms, 1664 x 16 4.1 ms, 1664 x 8 6.2 ms (two-phase entry: 6.0 ms).
			cap = 3072;
			const double m = 0.8 * 3072.0 / perCmd;
			segMax = m >= 64.0 ? 64u : (m >= 32.0 ? 32u : (m >= 16.0 ? 16u : 8u));
		}
	}
	if (ctx->optF1Cap) { cap = ctx->optF1Cap; }
	if (ctx->optF1Seg) { segMax = (uint32_t)ctx->optF1Seg; }
	// Very short curves (a handful of segments per cubic: a chunk of 64 commands yields a few hundred vertices) are bound by what
	// a SEGMENT costs in the one-walk kernel -- its dependent loads and its place in the order, ~40 000 cycles whatever it holds --
	// not by the walk: the two-walk kernels (count -> scan over draws -> emit, vgx_flatten.hip) are faster there (1 M cubics of
	// ~5 segments: 0.43 against 0.63 ms), and they need no host round trip either once the batch's command total is known from
	// the last call. Same output.
	// The per-command words of the two walks are sized for ndraws x (longest path): what ANY draw list of this size can need on this path
	// set (the tag only says that the set and the number of draws are the last call's -- the draws' paths may have changed).
	// A set whose longest path is far above its mean (the bound above twice the last call's total) takes the one-walk route instead.
	const uint64_t cmdBoundAll = ndraws * (uint64_t)(ps->maxCmdsPerPath ? ps->maxCmdsPerPath : 1);
	if (ctx->hostF1 && ctx->f1Tag == tag && ctx->hostF1[1] != 0 && ctx->hostF1[2] == tag && !ctx->optF1Cap && !ctx->optF1Seg && !ps->hasEmpty
		&& (double)ctx->hostF1[0] / (double)ctx->hostF1[1] * 64.0 <= 384.0 && cmdBoundAll <= 2 * ctx->hostF1[1] + 4096) {
		const uint64_t ncmdInst = cmdBoundAll;
		if ((st = ensure(ctx, ctx->cmdCnt, (ncmdInst + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
		// The device-side guard of THIS call only (a batch that grew: VGX_E_NOSPACE); the context's persistent cap -- what
		// vgx_tessellate relies on for the scratch sized by the last count -- is put back right after the launch.
		const uint64_t savedCmdCap = ctx->caps.cmd_instances;
		ctx->caps.cmd_instances = ctx->cmdCnt.cap / sizeof(uint32_t) - 1;
		// (the two-walk kernels write per-command words only -- no sub-path records: cmdCnt alone bounds this launch)
		runCmdPrefix(ctx, ps, draws, ndraws, s);
		ctx->caps.cmd_instances = savedCmdCap;
		const VgxCaps saved = ctx->caps;
		ctx->caps.poly_vertices = out->cap_poly_vertices; ctx->caps.subpaths = out->cap_subpaths; ctx->caps.meshes = ~0ull;
		runFlattenCount(ctx, ps, draws, ndraws, s); // the scan over the draws compares the totals with the caller's capacities
		ctx->caps = saved;
		VgxFlattenArgs a2 = flattenArgs(ctx, ps, draws, ndraws, apply_transform);
		a2.poly = out->poly; a2.subs = out->subpaths; a2.mdesc = nullptr; a2.mtab = nullptr;
		a2.caps.poly_vertices = ~0ull; a2.caps.subpaths = ~0ull; a2.caps.meshes = ~0ull;
		vgx_launch_flatten(true, a2, VGX_GRID_BLOCKS, s);
		mark(ctx, s, "flatten_emit");
		hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, s, (const VgxTotals*)ctx->totals.p, dev_sizes, dev_status);
		{
			VgxTotals* T = (VgxTotals*)ctx->totals.p;
			noteHip(ctx, hipMemcpyAsync(&ctx->hostF1[0], &T->sizes.num_poly_vertices, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
			noteHip(ctx, hipMemcpyAsync(&ctx->hostF1[1], &T->sizes.num_cmd_instances, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
		}
		if (out->draw_info && ndraws) {
			HIPCHK(ctx, hipMemcpyAsync(out->draw_info, ctx->dinfo.p, ndraws * sizeof(vgx_draw_info), hipMemcpyDeviceToDevice, s));
		}
		return launchStatus(ctx);
	}
	// Segments: buckets of at least min(32, segMax) command instances. The command total is not known on the host without a round
	// trip; ndraws x (longest path) bounds it (a batch whose bound is absurdly far above its real size asks once, synchronously).
	const uint64_t minItems = segMax < 32 ? segMax : 32;
	uint64_t cmdBound = ndraws * (uint64_t)(ps->maxCmdsPerPath ? ps->maxCmdsPerPath : 1);
	{
		const uint64_t savedCmdCap = ctx->caps.cmd_instances; // k_flat1 needs no per-command scratch: no guard for this launch, the
		ctx->caps.cmd_instances = ~0ull;                       // persistent cap (vgx_tessellate's guard) stays what the last count made it
		runCmdPrefix(ctx, ps, draws, ndraws, s);
		ctx->caps.cmd_instances = savedCmdCap;
	}
	if (cmdBound / minItems > (1ull << 26)) {
		if ((st = readTotals(ctx, s)) != VGX_OK) { return st; }
		if (ctx->hostTotals->status != VGX_OK) { return (int)ctx->hostTotals->status; }
		cmdBound = ctx->hostTotals->sizes.num_cmd_instances;
	}
	const uint64_t segBound = cmdBound / minItems + 2;
	if ((st = ensure(ctx, ctx->f1SegDraw, (segBound + 1) * sizeof(uint64_t))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->f1Segs, (segBound + segBound / 64 + 2) * sizeof(VgxF1Seg))) != VGX_OK) { return st; }
	VgxFlattenArgs a = flattenArgs(ctx, ps, draws, ndraws, apply_transform);
	a.poly = out->poly;
	a.subs = out->subpaths;
	a.mdesc = nullptr; a.mtab = nullptr;
	a.caps.poly_vertices = ~0ull; a.caps.subpaths = ~0ull; a.caps.meshes = ~0ull;
	VgxF1Args x;
	x.seg_draw = (uint64_t*)ctx->f1SegDraw.p; x.segs = (VgxF1Seg*)ctx->f1Segs.p; x.grps = x.segs + segBound;
	x.cap_poly = out->cap_poly_vertices; x.cap_subs = out->cap_subpaths; x.pass = 0; x.read_flags = 0; x.has_empty = ps->hasEmpty ? 1 : 0; x.seg_max = segMax; x.tag = tag;
	if (ps->hasSerial) { // statically serial paths: the exact builder counts their draws first (it marks them in dinfo)
		noteHip(ctx, hipMemsetAsync(ctx->dinfo.p, 0, ndraws * sizeof(vgx_draw_info), s));
		vgx_launch_flatten_serial(false, a, s);
	}
	const int waves = ctx->optF1Waves ? ctx->optF1Waves : 2048;
	vgx_launch_flat1(a, x, waves, cap, ps->hasSerial, s);
	// draws of the exact builder: every draw is inspected when the set has serial paths, else only the draws the kernel listed
	VgxFlattenArgs ae = a;
	ae.build_mode = ps->hasSerial ? 0 : 1;
	vgx_launch_flatten_serial(true, ae, s);
	vgx_launch_flat1_publish(a, x, dev_sizes, dev_status, s);
	mark(ctx, s, "flatten_one_walk");
	if (ctx->hostF1) { // for the next call's choice of instance / bucket (never waited for)
		VgxTotals* T = (VgxTotals*)ctx->totals.p;
		ctx->f1Tag = tag;
		noteHip(ctx, hipMemcpyAsync(&ctx->hostF1[0], &T->sizes.num_poly_vertices, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
		noteHip(ctx, hipMemcpyAsync(&ctx->hostF1[1], &T->sizes.num_cmd_instances, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
		noteHip(ctx, hipMemcpyAsync(&ctx->hostF1[2], &T->flat_tag, sizeof(unsigned long long), hipMemcpyDeviceToHost, s)); // the tag lands behind the two values
	}
	if (out->draw_info && ndraws) {
		HIPCHK(ctx, hipMemcpyAsync(out->draw_info, ctx->dinfo.p, ndraws * sizeof(vgx_draw_info), hipMemcpyDeviceToDevice, s));
	}
	return launchStatus(ctx);
}

// ---- partition (contiguous ranges per GPU, balanced on the count pass for heterogeneous batches) -------------
namespace {
// cut k of nparts: the first draw whose weight prefix reaches k / nparts of the total; `align` > 1 (the draws repeat a sequence of
// `align` paths: a drawing submitted for many instances): rounded to the nearest whole instance, so that every part stays a batch
// of whole instances (what the instanced / template paths of vgx_tessellate need)
__device__ __forceinline__ uint64_t partition_cut(const uint64_t* prefix, uint64_t ndraws, uint32_t nparts, uint32_t k, uint64_t align)
{
	if (k >= nparts) { return ndraws; }
	const uint64_t total = prefix[ndraws];
	const uint64_t target = (uint64_t)(((unsigned __int128)total * k) / nparts);
	uint64_t lo = 0, hi = ndraws;
	while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (prefix[mid] < target) { lo = mid + 1; } else { hi = mid; } }
	if (align > 1) { lo = (lo + align / 2) / align * align; if (lo > ndraws) { lo = ndraws; } }
	return lo;
}
__global__ void k_partition_bounds(const uint64_t* prefix, uint64_t ndraws, uint32_t nparts, uint64_t align, uint64_t* bounds, uint64_t* weights)
{
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k > nparts) { return; }
	const uint64_t b = partition_cut(prefix, ndraws, nparts, k, align);
	bounds[k] = b;
	if (weights && k < nparts) { weights[k] = prefix[partition_cut(prefix, ndraws, nparts, k + 1, align)] - prefix[b]; }
}
}

int vgx_partition(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, uint32_t nparts, uint64_t* out_bounds, uint64_t* out_weights, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !ps || (!draws && ndraws) || !out_bounds || nparts == 0 || nparts > 65536u) {
		return VGX_E_INVALID_ARG;
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	ctx->lastStage = 0;
	int st;
	uint64_t align = 1;
	if (ctx->optInst && ndraws > VGX_SMALL_DRAWS) { // a drawing submitted for many instances? Then the cuts fall between instances
		if ((st = ensure(ctx, ctx->totals, sizeof(VgxTotals))) != VGX_OK) { return st; }
		noteHip(ctx, hipMemsetAsync(ctx->totals.p, 0, sizeof(VgxTotals), s));
		vgx_launch_inst_detect(draws, ndraws, (VgxTotals*)ctx->totals.p, s);
		if ((st = readTotals(ctx, s)) != VGX_OK) { return st; }
		if (ctx->hostTotals->inst_detect_inv != 0 && !ctx->hostTotals->inst_detect_bad) {
			const unsigned long long P = ~0ull - ctx->hostTotals->inst_detect_inv;
			if (P > 1 && ndraws % P == 0 && ndraws / P >= nparts) { align = P; }
		}
	}
	st = flattenCountCommon(ctx, ps, draws, ndraws, s, false); // per-draw polyline vertex counts in dinfo
	if (st != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->partBounds, ((size_t)nparts + 1) * 2 * sizeof(uint64_t))) != VGX_OK) { return st; }
	OpPartWeight op;
	op.draws = draws; op.dinfo = (const vgx_draw_info*)ctx->dinfo.p; op.ndraws = ndraws; op.prefix = (uint64_t*)ctx->cmdPrefix.p; // the command prefix is not needed any more
	vgx_device_scan(op, (Sum3*)ctx->partial.p, s, ndraws);
	uint64_t* dBounds = (uint64_t*)ctx->partBounds.p;
	uint64_t* dWeights = dBounds + nparts + 1;
	hipLaunchKernelGGL(k_partition_bounds, dim3((nparts + 1 + 255) / 256), dim3(256), 0, s, (const uint64_t*)ctx->cmdPrefix.p, ndraws, nparts, align, dBounds, dWeights);
	HIPCHK(ctx, hipMemcpyAsync(out_bounds, dBounds, ((size_t)nparts + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	if (out_weights) { HIPCHK(ctx, hipMemcpyAsync(out_weights, dWeights, (size_t)nparts * sizeof(uint64_t), hipMemcpyDeviceToHost, s)); }
	HIPCHK(ctx, hipStreamSynchronize(s));
	return VGX_OK;
}

// ---- template mode (vgx_tmpl.hip) ---------------------------------------------------------------------
// One step in template mode: verify every draw against the saved first period, emit. No scratch besides the template.
// Round-join templates: the per-step tables (sized for this batch) and the kernels that fill them -- sizes of every Round-join mesh of
// every instance, places of meshes and instances, the batch totals in totals->sizes, VGX_E_NOSPACE against a.caps.
static int tmplRoundSizes(vgx_ctx* ctx, VgxTmplArgs& a, hipStream_t s)
{
	int st;
	const uint64_t n = a.ninst;
	a.num_round = ctx->tmplRound; a.num_round_elems = ctx->tmplRoundElems;
	a.trmesh = (const VgxTmplRoundMesh*)ctx->tmplTrmesh.p; a.tmsz = (const uint2*)ctx->tmplTmsz.p;
	if (n * a.num_round >= (1ull << 31)) { return VGX_E_RANGE; } // one wave (long meshes: one workgroup) per (instance, Round-join mesh)
	const bool classes = a.cls != nullptr; // several classes: the tables' rows are the instances' own (VgxTmplInst::m / ::rel), the sizes stay in LDS
	if (!classes && (st = ensure(ctx, ctx->tmplRsz, (n * a.num_round + 1) * 2 * sizeof(unsigned long long))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->tmplRelem, ((classes ? ctx->tmplRelemWords : n * a.num_round_elems) + 1) * sizeof(uint2))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->tmplMplace, ((classes ? a.total.num_meshes : n * a.inst.num_meshes) + 1) * sizeof(VgxTmplMeshPlace))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->partial, VGX_SCAN_BLOCKS * sizeof(Sum3))) != VGX_OK) { return st; }
	a.rsz = classes ? nullptr : (unsigned long long*)ctx->tmplRsz.p; a.relem = (uint2*)ctx->tmplRelem.p;
	a.mplace = (VgxTmplMeshPlace*)ctx->tmplMplace.p;
	if (vgx_tmpl_round_per_instance(a)) {
		if ((st = ensure(ctx, ctx->tmplItot, (n + 1) * 2 * sizeof(unsigned long long))) != VGX_OK) { return st; }
		if ((st = ensure(ctx, ctx->tmplIplace, (n + 1) * 2 * sizeof(unsigned long long))) != VGX_OK) { return st; }
		a.itot = (unsigned long long*)ctx->tmplItot.p; a.iplace = (unsigned long long*)ctx->tmplIplace.p;
	}
	vgx_launch_tmpl_round_sizes(a, (Sum3*)ctx->partial.p, s);
	mark(ctx, s, "tmpl_round_sizes");
	return VGX_OK;
}

// the step's arguments but the caller's buffers
static void tmplArgs(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, VgxTmplArgs& a)
{
	a.draws = draws; a.ndraws = ndraws; a.period = ctx->tmplPeriod; a.ninst = ndraws / ctx->tmplPeriod; a.npaths = ps->dev.npaths;
	a.tdraws = (const vgx_draw*)ctx->tmplDraws.p; a.tpoly = (const float2*)ctx->tmplPoly.p; a.tmesh = (const VgxTmplMesh*)ctx->tmplMesh.p;
	a.tmtab = (const vgx_mesh*)ctx->tmplMtab.p; a.telem = (const VgxTmplElem*)ctx->tmplElem.p;
	a.inst = ctx->tmplInst;
	a.ttile = (const VgxTmplTile*)ctx->tmplTile.p;
	a.tile = ctx->tmplTileSize;
	a.tiles_per_inst = (uint32_t)((ctx->tmplInst.num_elements + a.tile - 1) / a.tile);
	a.caps = ctx->caps;
	a.totals = (VgxTotals*)ctx->totals.p;
	a.general = ctx->tmplGeneral;
	if (ctx->tmplClasses > 1) {
		a.iinfo = (const VgxTmplInst*)ctx->tmplIinfo.p; a.wg = (const uint2*)ctx->tmplWg.p; a.num_wg = ctx->tmplNumWg;
		a.cls = (const VgxTmplClass*)ctx->tmplCls.p; a.round_lds = ctx->tmplRoundLds;
		a.total = ctx->tmplTotal;
	} else {
		const vgx_sizes& i1 = ctx->tmplInst;
		vgx_sizes z;
		z.num_poly_vertices = a.ninst * i1.num_poly_vertices; z.num_subpaths = a.ninst * i1.num_subpaths; z.num_meshes = a.ninst * i1.num_meshes;
		z.num_vertices = a.ninst * i1.num_vertices; z.num_indices = a.ninst * i1.num_indices; z.num_serial_draws = a.ninst * i1.num_serial_draws;
		z.num_cmd_instances = a.ninst * i1.num_cmd_instances; z.num_elements = a.ninst * i1.num_elements; z.num_fill_elements = a.ninst * i1.num_fill_elements;
		z.num_drawcmds = 0;
		a.total = z;
		a.num_wg = a.ninst * (uint64_t)a.tiles_per_inst;
	}
}

static int runTmpl(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, hipStream_t s)
{
	VgxTmplArgs a;
	memset(&a, 0, sizeof(a));
	tmplArgs(ctx, ps, draws, ndraws, a);
	a.pos = out->pos; a.color = out->color; a.idx = out->idx; a.meshes_out = out->meshes;
	a.caps.vertices = out->cap_vertices; a.caps.indices = out->cap_indices; a.caps.meshes = out->cap_meshes;
	if (a.num_wg > 0x7FFFFFFFull) { return VGX_E_RANGE; } // one workgroup per (instance, tile)
	noteHip(ctx, hipMemsetAsync(ctx->totals.p, 0, sizeof(VgxTotals), s));
	if (a.general == 3 || a.general == 5 || a.general == 6) {
		// Round joins: vertices / indices are the instances' own -- counted on the device, checked against the capacities there
		a.total.num_vertices = 0; a.total.num_indices = 0;
		int st;
		if ((st = tmplRoundSizes(ctx, a, s)) != VGX_OK) { return st; }
	} else {
		// every size is known on the host: a batch that does not fit the caller's buffers ends here, with the need in dev_sizes
		const uint64_t nv = a.total.num_vertices, ni = a.total.num_indices, nm = a.total.num_meshes;
		const uint32_t aux = (nv > out->cap_vertices ? 1u : 0u) | (ni > out->cap_indices ? 2u : 0u) | ((out->meshes && nm > out->cap_meshes) ? 4u : 0u);
		if (aux) {
			const vgx_sizes z = a.total;
			hipLaunchKernelGGL(k_tmpl_nospace, dim3(1), dim3(1), 0, s, (VgxTotals*)ctx->totals.p, z, aux);
			if (dev_sizes || dev_status) { hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, s, (const VgxTotals*)ctx->totals.p, dev_sizes, dev_status); }
			return launchStatus(ctx);
		}
	}
	if (ctx->asmArmed) {
		// draw-command assembly: the partition kernels need the whole batch's mesh table and every mesh's draw in memory
		const uint64_t nm = a.total.num_meshes;
		int st;
		if ((st = ensure(ctx, ctx->mtab, (nm + 1) * sizeof(vgx_mesh))) != VGX_OK) { return st; }
		if ((st = ensure(ctx, ctx->mdesc, (nm + 1) * sizeof(VgxMeshDesc))) != VGX_OK) { return st; }
		vgx_launch_tmpl_mtab(a, (vgx_mesh*)ctx->mtab.p, (VgxMeshDesc*)ctx->mdesc.p, s);
		mark(ctx, s, "tmpl_mesh_table");
		if ((st = runAssemble(ctx, out, s, draws)) != VGX_OK) { return st; }
		a.mesh_base = (const uint32_t*)ctx->meshBase.p;
	}
	vgx_launch_tmpl_emit(a, s);
	mark(ctx, s, "tmpl_emit");
	if (dev_sizes || dev_status) {
		hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, s, (const VgxTotals*)ctx->totals.p, dev_sizes, dev_status);
	}
	return launchStatus(ctx);
}

static bool tmplFor(const vgx_ctx* ctx, const vgx_pathset* ps, uint64_t ndraws)
{
	return ctx->tmplOn && ctx->optTmpl && ps == ctx->tmplPs && ps->gen == ctx->tmplPsGen && ctx->tmplPeriod && ndraws % ctx->tmplPeriod == 0 && ndraws >= ctx->tmplPeriod
		&& (ctx->tmplClasses == 1 || ndraws == ctx->tmplNDraws); // several classes: the per-instance table belongs to ONE batch size
}

// The ordinary count + two-phase flatten in LOCAL space (apply_transform = 0) + mesh sizing of `n` draws: what a template is built
// from. Sizes and the mesh-kind flags end up in ctx->hostTotals.
static int tmplPipeline(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* d, uint64_t n, hipStream_t s)
{
	int st;
	if ((st = flattenCountCommon(ctx, ps, d, n, s, false)) != VGX_OK) { return st; }
	const vgx_sizes fsz = ctx->hostTotals->sizes;
	if ((st = ensureMeshBuffers(ctx, fsz.num_poly_vertices, fsz.num_subpaths, fsz.num_meshes)) != VGX_OK) { return st; }
	{
		VgxFlattenArgs a = flattenArgs(ctx, ps, d, n, 0);
		vgx_launch_flatten(true, a, VGX_GRID_BLOCKS, s);
	}
	VgxCaps outCaps = ctx->caps;
	outCaps.vertices = ~0ull; outCaps.indices = ~0ull;
	runStrokeCount(ctx, d, outCaps, 0, s);
	if ((st = readTotals(ctx, s)) != VGX_OK) { return st; }
	if (ctx->hostTotals->status != VGX_OK) { return (int)ctx->hostTotals->status; }
	return VGX_OK;
}
// one instance of a class: only the mesh kinds k_tmpl_emit writes, every output stream of the instance below 4 GB
static bool tmplEligible(const VgxTotals& ht, bool roundOk)
{
	const vgx_sizes& z = ht.sizes;
	return !((ht.num_round_meshes && !roundOk) || z.num_elements == 0 || z.num_elements >= (1ull << 31) || z.num_vertices >= (1ull << 29)
		|| z.num_indices >= (1ull << 31) || z.num_poly_vertices >= (1ull << 32) || z.num_meshes >= (1ull << 32));
}

// vgx_tessellate_count, first thing: do the draws repeat their first period in everything but transform and colours -- or a FEW
// flavours of it ("classes": every instance equals one of at most VGX_TMPL_MAX_CLASSES representatives, e.g. one drawing at a
// handful of scales) --, and are the meshes all of the kinds k_tmpl_emit writes (fills; closed Miter AA / Thin strokes)? Then the
// representatives are flattened ONCE, in local space, by the ordinary two-phase kernels and kept as the context's template.
// Returns VGX_OK with ctx->tmplOn set (out_sizes filled), VGX_OK with it clear (not such a batch: the caller continues with the
// ordinary count), or an error.
static int tryTemplate(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, vgx_sizes* out_sizes, hipStream_t s)
{
	ctx->tmplOn = false;
	if (!ctx->optTmpl || !ctx->optInst || ctx->optTwoPass || (ndraws <= VGX_SMALL_DRAWS && !ctx->optTmplBatch) || ndraws == 0) { return VGX_OK; } // (static batches: frame-sized draw lists too)
	int st;
	if ((st = ensure(ctx, ctx->totals, sizeof(VgxTotals))) != VGX_OK) { return st; }
	noteHip(ctx, hipMemsetAsync(ctx->totals.p, 0, sizeof(VgxTotals), s));
	vgx_launch_inst_detect(draws, ndraws, (VgxTotals*)ctx->totals.p, s);
	vgx_launch_tmpl_check(draws, ndraws, ps->dev.npaths, (VgxTotals*)ctx->totals.p, s);
	if ((st = readTotals(ctx, s)) != VGX_OK) { return st; }
	if (ctx->hostTotals->status != VGX_OK) { return (int)ctx->hostTotals->status; }
	ctx->tmplIsBatch = false;
	unsigned long long P = (ctx->hostTotals->inst_detect_inv == 0 || ctx->hostTotals->inst_detect_bad) ? 0ull : ~0ull - ctx->hostTotals->inst_detect_inv;
	if (P == 0 || P > (1ull << 24) || ndraws % P != 0 || ndraws / P < VGX_INST_MIN_INSTANCES) {
		// no drawing repeated for many instances. A caller that keeps its batches' STRUCTURE from call to call (vgx_set_static_batches: the
		// same paths and styles in the same order, only transforms and colours move -- a static scene under a moving camera, a culled or
		// shuffled instanced scene) gets the whole draw list as ONE template of one instance: flattened once here, in local space; a step is
		// then the emit kernel alone, and any structural change ends it with VGX_E_STALE (the caller counts again)
		if (!ctx->optTmplBatch || ndraws >= (1ull << 31)) { return VGX_OK; }
		P = ndraws;
		ctx->tmplIsBatch = true;
		ctx->hostTotals->tmpl_bad = 0;
	}
	const uint64_t ninst = ndraws / P;
	uint32_t T = 1;
	std::vector<uint32_t> instCls, reps(1, 0u);
	if (ctx->hostTotals->tmpl_bad) {
		// not ONE period repeated. A few flavours of it? Candidates by hash of the template fields, then compared bit by bit.
		if (!ctx->optTmplClasses || ninst > (1ull << 24)) { return VGX_OK; } // (per-instance tables: 40 bytes per instance and tile on host and device)
		if ((st = ensure(ctx, ctx->tmplHash, ninst * sizeof(unsigned long long))) != VGX_OK) { return st; }
		noteHip(ctx, hipMemsetAsync(ctx->tmplHash.p, 0, ninst * sizeof(unsigned long long), s));
		vgx_launch_tmpl_hash(draws, ndraws, P, (unsigned long long*)ctx->tmplHash.p, s);
		std::vector<unsigned long long> hashes(ninst);
		HIPCHK(ctx, hipMemcpyAsync(hashes.data(), ctx->tmplHash.p, ninst * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
		HIPCHK(ctx, hipStreamSynchronize(s));
		instCls.resize(ninst);
		reps.clear();
		std::unordered_map<unsigned long long, uint32_t> seen;
		for (uint64_t k = 0; k < ninst; ++k) {
			const auto it = seen.find(hashes[k]);
			uint32_t c;
			if (it != seen.end()) { c = it->second; }
			else {
				c = (uint32_t)reps.size();
				if (c == VGX_TMPL_MAX_CLASSES) { return VGX_OK; } // too many flavours: the ordinary pipeline
				seen.emplace(hashes[k], c);
				reps.push_back((uint32_t)k);
			}
			instCls[k] = c;
		}
		T = (uint32_t)reps.size();
		if ((st = ensure(ctx, ctx->tmplInstCls, ninst * sizeof(uint32_t))) != VGX_OK) { return st; }
		if ((st = ensure(ctx, ctx->tmplClsRep, (size_t)T * sizeof(uint32_t))) != VGX_OK) { return st; }
		HIPCHK(ctx, hipMemcpyAsync(ctx->tmplInstCls.p, instCls.data(), ninst * sizeof(uint32_t), hipMemcpyHostToDevice, s));
		HIPCHK(ctx, hipMemcpyAsync(ctx->tmplClsRep.p, reps.data(), (size_t)T * sizeof(uint32_t), hipMemcpyHostToDevice, s));
		noteHip(ctx, hipMemsetAsync(&((VgxTotals*)ctx->totals.p)->tmpl_bad, 0, sizeof(uint32_t), s));
		vgx_launch_tmpl_check_cls(draws, ndraws, P, (const uint32_t*)ctx->tmplInstCls.p, (const uint32_t*)ctx->tmplClsRep.p, (VgxTotals*)ctx->totals.p, s);
		if ((st = readTotals(ctx, s)) != VGX_OK) { return st; }
		if (ctx->hostTotals->tmpl_bad) { return VGX_OK; } // equal hashes, different records
	}
	// the class representatives' draw records, back to back (the saved records every step verifies the batch against)
	const uint64_t PT = P * T;
	if ((st = ensure(ctx, ctx->tmplDraws, (size_t)PT * sizeof(vgx_draw))) != VGX_OK) { return st; }
	for (uint32_t c = 0; c < T; ++c) {
		HIPCHK(ctx, hipMemcpyAsync((vgx_draw*)ctx->tmplDraws.p + (uint64_t)c * P, draws + (uint64_t)reps[c] * P, (size_t)P * sizeof(vgx_draw), hipMemcpyDeviceToDevice, s));
	}
	const vgx_draw* rdraws = (const vgx_draw*)ctx->tmplDraws.p;
	// sizes of one instance of every class (several classes: each representative through the pipeline on its own first)
	// (several classes, round 6: NOT one pipeline per representative -- their sizes are differences of the prefixes the one concatenated
	// run leaves behind, k_tmpl_class_sums below; the Tiger at seven scales = 18 classes: count 4.2 -> ~1 ms)
	std::vector<vgx_sizes> csz(T);
	// all representatives as one batch: the template
	if ((st = tmplPipeline(ctx, ps, rdraws, PT, s)) != VGX_OK) { return st; }
	{
		const VgxTotals& ht = *ctx->hostTotals;
		if (T == 1) {
			// Round joins (their sizes are the instance's): templates of one class; else the ordinary pipeline
			if (!tmplEligible(ht, ctx->optTmplRound != 0)) { return VGX_OK; }
			csz[0] = ht.sizes;
		} else if ((ht.num_round_meshes && !ctx->optTmplRound) || ht.sizes.num_poly_vertices >= (1ull << 32) || ht.sizes.num_meshes >= (1ull << 32) || ht.sizes.num_elements >= (1ull << 36)) {
			return VGX_OK;
		}
	}
	const vgx_sizes all = ctx->hostTotals->sizes;
	const bool general = ctx->hostTotals->has_general_stroke != 0;
	const uint64_t M = all.num_meshes, E = all.num_elements, V = all.num_poly_vertices;
	if ((st = ensure(ctx, ctx->tmplCls, ((size_t)T + 1) * sizeof(VgxTmplClass))) != VGX_OK) { return st; }
	VgxTmplBuild b;
	memset(&b, 0, sizeof(b));
	// which stroke styles does the template hold? (decides the emit kernel and, with it, the tile size)
	uint32_t styles = 0;
	if (general) {
		b.mdesc = (const VgxMeshDesc*)ctx->mdesc.p; b.num_meshes = M; b.nclasses = T; b.cls = (VgxTmplClass*)ctx->tmplCls.p;
		noteHip(ctx, hipMemsetAsync(ctx->tmplCls.p, 0, ((size_t)T + 1) * sizeof(VgxTmplClass), s));
		vgx_launch_tmpl_styles(b, s);
		HIPCHK(ctx, hipMemcpyAsync(&styles, &((VgxTmplClass*)ctx->tmplCls.p)[T].pad[0], sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		HIPCHK(ctx, hipStreamSynchronize(s));
	} else {
		noteHip(ctx, hipMemsetAsync(ctx->tmplCls.p, 0, ((size_t)T + 1) * sizeof(VgxTmplClass), s));
	}
	// (bit 3: closed Bevel strokes -- a kernel of their own beside the closed Miter ones; with open strokes or anything else in the template, the general one)
	// (bit 4: closed AA strokes with Round joins -- with nothing but closed strokes in the template, kernel 5: no general body)
	const uint32_t kernelKind = (styles & 4u) ? ((styles & 3u) ? 3u : ((styles & 32u) ? 6u : 5u)) : ((styles & 2u) ? 2u : ((styles & 8u) ? ((styles & 1u) ? 2u : 4u) : ((styles & 1u) ? 1u : 0u)));
	const bool roundTmpl = kernelKind == 3u || kernelKind == 5u || kernelKind == 6u;
	// (Round joins in a template of several classes, round 6: the per-step tables are addressed per instance -- VgxTmplInst::m / ::rel --, the
	// sizes pass runs in its workgroup-per-instance shape; a batch that does not fit that shape takes the ordinary pipeline, see below)
	uint32_t tileSize = ((kernelKind == 2u || kernelKind == 3u) && ctx->optTmplTile > VGX_TMPL_GENERAL_TILE) ? (uint32_t)VGX_TMPL_GENERAL_TILE : ctx->optTmplTile;
	if ((kernelKind == 5u || kernelKind == 6u) && ctx->optTmplTile == VGX_TMPL_MAX_TILE) { tileSize = VGX_TMPL_RC_TILE; } // (its own workgroup shape; a VGX_TMPL_TILE override stands)
	b.draws = rdraws; b.poly = (const float2*)ctx->poly.p; b.mdesc = (const VgxMeshDesc*)ctx->mdesc.p; b.mprep = (const VgxMeshPrep*)ctx->mprep.p; b.mtab = (const vgx_mesh*)ctx->mtab.p;
	b.prefix_fill = (const uint64_t*)ctx->elemPrefix.p; b.prefix_stroke = (const uint64_t*)ctx->elemPrefixS.p;
	b.num_meshes = M; b.num_elems = E; b.tile = tileSize; b.period = (uint32_t)P; b.nclasses = T;
	b.num_vertices = all.num_vertices; b.num_indices = all.num_indices; b.cls = (VgxTmplClass*)ctx->tmplCls.p;
	vgx_launch_tmpl_classes(b, s);
	std::vector<VgxTmplClass> cls((size_t)T + 1);
	std::vector<unsigned long long> csum(((size_t)T + 1) * 5, 0ull);
	if (T > 1) {
		if ((st = ensure(ctx, ctx->tmplClsSum, ((size_t)T + 1) * 5 * sizeof(unsigned long long))) != VGX_OK) { return st; }
		vgx_launch_tmpl_class_sums(b, (const vgx_draw_info*)ctx->dinfo.p, (const uint64_t*)ctx->cmdPrefix.p, PT, all, (unsigned long long*)ctx->tmplClsSum.p, s);
		HIPCHK(ctx, hipMemcpyAsync(csum.data(), ctx->tmplClsSum.p, csum.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
	}
	HIPCHK(ctx, hipMemcpyAsync(cls.data(), ctx->tmplCls.p, ((size_t)T + 1) * sizeof(VgxTmplClass), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	if (T > 1) { // sizes of ONE instance of every class = what the concatenated run holds between the class's first draw and the next class's
		for (uint32_t c = 0; c < T; ++c) {
			vgx_sizes q;
			memset(&q, 0, sizeof(q));
			q.num_meshes = cls[c + 1].mesh0 - cls[c].mesh0; q.num_elements = cls[c + 1].elem0 - cls[c].elem0;
			q.num_vertices = cls[c + 1].v0 - cls[c].v0; q.num_indices = cls[c + 1].i0 - cls[c].i0;
			const unsigned long long* a0 = &csum[(size_t)c * 5]; const unsigned long long* a1 = &csum[(size_t)(c + 1) * 5];
			q.num_poly_vertices = a1[0] - a0[0]; q.num_subpaths = a1[1] - a0[1]; q.num_cmd_instances = a1[2] - a0[2];
			q.num_fill_elements = a1[3] - a0[3]; q.num_serial_draws = a1[4]; // (entry c + 1 holds class c's own count)
			csz[c] = q;
			VgxTotals one; // the limits of one instance (tmplEligible), per class as before
			memset(&one, 0, sizeof(one));
			one.sizes = q;
			if (!tmplEligible(one, false)) { return VGX_OK; }
		}
	}
	const uint64_t tiles = cls[T].tile0;
	if ((st = ensure(ctx, ctx->tmplPoly, (V + 1) * 2 * sizeof(float))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->tmplMesh, (M + 1) * sizeof(VgxTmplMesh))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->tmplMtab, (M + 1) * sizeof(vgx_mesh))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->tmplElem, (tiles * tileSize + 64) * sizeof(VgxTmplElem))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->tmplTile, (tiles + 1) * sizeof(VgxTmplTile))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->tmplTrmesh, (M + 2) * sizeof(VgxTmplRoundMesh))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->tmplTmsz, (M + 1) * sizeof(uint2))) != VGX_OK) { return st; }
	b.tmsz = (uint2*)ctx->tmplTmsz.p;
	b.trmesh = (VgxTmplRoundMesh*)ctx->tmplTrmesh.p;
	b.has_round = roundTmpl ? 1u : 0u;
	if ((st = ensure(ctx, ctx->partial, VGX_SCAN_BLOCKS * sizeof(Sum3))) != VGX_OK) { return st; }
	b.partial = (Sum3*)ctx->partial.p;
	b.ttile = (VgxTmplTile*)ctx->tmplTile.p;
	b.tmesh = (VgxTmplMesh*)ctx->tmplMesh.p; b.tmtab = (vgx_mesh*)ctx->tmplMtab.p; b.telem = (VgxTmplElem*)ctx->tmplElem.p;
	vgx_launch_tmpl_build(b, s);
	HIPCHK(ctx, hipMemcpyAsync(ctx->tmplPoly.p, ctx->poly.p, V * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
	// Round joins: how many such meshes the template holds and their elements; several classes: where each class's begin (k_tmpl_round_classes)
	uint32_t roundWord = 0, roundElems = 0, roundLds = 0;
	std::vector<uint64_t> clsRelems(T, 0); // Round-join elements of one instance of every class
	if (roundTmpl) {
		HIPCHK(ctx, hipMemcpyAsync(cls.data(), ctx->tmplCls.p, ((size_t)T + 1) * sizeof(VgxTmplClass), hipMemcpyDeviceToHost, s));
		HIPCHK(ctx, hipStreamSynchronize(s));
		roundWord = cls[T].pad[1];
		if (roundWord == 0) { return VGX_OK; }
		VgxTmplRoundMesh last;
		HIPCHK(ctx, hipMemcpyAsync(&last, (const VgxTmplRoundMesh*)ctx->tmplTrmesh.p + roundWord, sizeof(last), hipMemcpyDeviceToHost, s));
		HIPCHK(ctx, hipStreamSynchronize(s));
		roundElems = last.elem0;
		roundLds = roundWord;
		clsRelems[0] = roundElems;
		if (T > 1) {
			roundLds = 0;
			for (uint32_t c = 0; c < T; ++c) {
				const uint32_t r1 = c + 1 < T ? cls[c + 1].pad[1] : roundWord, e1 = c + 1 < T ? cls[c + 1].pad[0] : roundElems;
				clsRelems[c] = e1 - cls[c].pad[0];
				if (r1 - cls[c].pad[1] > roundLds) { roundLds = r1 - cls[c].pad[1]; }
			}
		}
	}
	// the batch: instances x their class's sizes
	std::vector<uint64_t> cnt(T, 0);
	if (T == 1) { cnt[0] = ninst; } else { for (uint64_t k = 0; k < ninst; ++k) { ++cnt[instCls[k]]; } }
	vgx_sizes z;
	memset(&z, 0, sizeof(z));
	uint64_t numWg = 0;
	for (uint32_t c = 0; c < T; ++c) {
		const vgx_sizes& q = csz[c];
		z.num_poly_vertices += cnt[c] * q.num_poly_vertices; z.num_subpaths += cnt[c] * q.num_subpaths; z.num_meshes += cnt[c] * q.num_meshes;
		z.num_vertices += cnt[c] * q.num_vertices; z.num_indices += cnt[c] * q.num_indices; z.num_serial_draws += cnt[c] * q.num_serial_draws;
		z.num_cmd_instances += cnt[c] * q.num_cmd_instances; z.num_elements += cnt[c] * q.num_elements; z.num_fill_elements += cnt[c] * q.num_fill_elements;
		numWg += cnt[c] * (uint64_t)(cls[c + 1].tile0 - cls[c].tile0);
	}
	if (T > 1) {
		if (numWg > 0x7FFFFFFFull || z.num_meshes >= (1ull << 32)) { return VGX_OK; }
		std::vector<VgxTmplInst> ii(ninst + 1);
		std::vector<uint2> wg((size_t)numWg);
		uint64_t v = 0, i = 0, m = 0, w = 0, re = 0;
		for (uint64_t k = 0; k <= ninst; ++k) {
			VgxTmplInst r;
			memset(&r, 0, sizeof(r));
			r.v = v; r.i = i; r.m = (uint32_t)m;
			if (k < ninst) {
				const uint32_t c = instCls[k];
				r.cls = c; r.cmesh0 = cls[c].mesh0;
				r.rel = roundTmpl ? re - (uint64_t)cls[c].pad[0] : 0ull; // (the element numbers in the mesh records run over the whole template)
				v += csz[c].num_vertices; i += csz[c].num_indices; m += csz[c].num_meshes; re += clsRelems[c];
			}
			ii[(size_t)k] = r;
		}
		// Workgroup order: class after class (the instances of a class in draw order). The output places are the instances' own
		// whatever the order; running one class's instances together keeps ONE class's tables hot in L2 instead of all of them
		// (instances in draw order: 3.9 GB of table re-reads from HBM for Tiger x 10k in 18 classes).
		std::vector<uint64_t> cursor(T, 0);
		for (uint32_t c = 0; c < T; ++c) { cursor[c] = w; w += cnt[c] * (uint64_t)(cls[c + 1].tile0 - cls[c].tile0); }
		for (uint64_t k = 0; k < ninst; ++k) {
			const uint32_t c = instCls[k];
			for (uint32_t t = cls[c].tile0; t < cls[c + 1].tile0; ++t) { wg[(size_t)cursor[c]++] = make_uint2((uint32_t)k, t); }
		}
		if ((st = ensure(ctx, ctx->tmplIinfo, (ninst + 1) * sizeof(VgxTmplInst))) != VGX_OK) { return st; }
		if ((st = ensure(ctx, ctx->tmplWg, ((size_t)numWg + 1) * sizeof(uint2))) != VGX_OK) { return st; }
		HIPCHK(ctx, hipMemcpyAsync(ctx->tmplIinfo.p, ii.data(), (ninst + 1) * sizeof(VgxTmplInst), hipMemcpyHostToDevice, s));
		HIPCHK(ctx, hipMemcpyAsync(ctx->tmplWg.p, wg.data(), (size_t)numWg * sizeof(uint2), hipMemcpyHostToDevice, s));
		HIPCHK(ctx, hipStreamSynchronize(s)); // the host vectors go away
	}
	HIPCHK(ctx, hipStreamSynchronize(s));
	if ((st = launchStatus(ctx)) != VGX_OK) { return st; }
	ctx->tmplRound = roundWord;
	ctx->tmplRoundElems = roundElems;
	ctx->tmplRoundLds = roundLds;
	ctx->tmplRelemWords = 0;
	for (uint32_t c = 0; c < T; ++c) { ctx->tmplRelemWords += cnt[c] * clsRelems[c]; }
	ctx->tmplInst = csz[0];
	ctx->tmplTileSize = tileSize;
	ctx->tmplPeriod = (uint32_t)P;
	ctx->tmplPs = ps;
	ctx->tmplPsGen = ps->gen;
	ctx->tmplClasses = T;
	ctx->tmplNumWg = numWg;
	ctx->tmplNDraws = ndraws;
	ctx->tmplTotal = z;
	ctx->tmplGeneral = kernelKind; // which instantiation of the emit kernel the template needs
	if (roundTmpl) {
		// Round joins: this batch's vertices / indices -- the count of one step, read back
		VgxTmplArgs a;
		memset(&a, 0, sizeof(a));
		tmplArgs(ctx, ps, draws, ndraws, a);
		a.caps.vertices = ~0ull; a.caps.indices = ~0ull; a.caps.meshes = ~0ull;
		a.total.num_vertices = 0; a.total.num_indices = 0;
		if (a.num_wg > 0x7FFFFFFFull || a.ninst * (uint64_t)roundWord >= (1ull << 31)) { return VGX_OK; } // beyond the sizes kernels' grids: the ordinary pipeline
		if (T > 1) { // several classes: only the workgroup-per-instance shape of the sizes pass addresses its tables per instance
			a.num_round = roundWord; a.num_round_elems = roundElems;
			if (!vgx_tmpl_round_per_instance(a) || ctx->tmplRelemWords >= (1ull << 40)) { return VGX_OK; }
		}
		noteHip(ctx, hipMemsetAsync(ctx->totals.p, 0, sizeof(VgxTotals), s));
		if ((st = tmplRoundSizes(ctx, a, s)) != VGX_OK) { return st; }
		if ((st = readTotals(ctx, s)) != VGX_OK) { return st; }
		if (ctx->hostTotals->status != VGX_OK) { return (int)ctx->hostTotals->status; }
		z = ctx->hostTotals->sizes;
		ctx->tmplTotal = z;
	}
	ctx->tmplOn = true;
	*out_sizes = z;
	ctx->hostTotals->sizes = z; // what vgx_tessellate_emit checks the caller's capacities against
	return VGX_OK;
}

// ---- tessellate ---------------------------------------------------------------------------------------
int vgx_tessellate_count(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, vgx_sizes* out_sizes, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !ps || (!draws && ndraws) || !out_sizes) {
		return VGX_E_INVALID_ARG;
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	ctx->lastStage = 0;
	int st = tryTemplate(ctx, ps, draws, ndraws, out_sizes, s);
	if (st != VGX_OK) { return st; }
	if (ctx->tmplOn) {
		ctx->lastPs = ps; ctx->lastDraws = draws; ctx->lastNDraws = ndraws; ctx->lastStage = 2;
		return VGX_OK;
	}
	st = flattenCountCommon(ctx, ps, draws, ndraws, s, true);
	if (st != VGX_OK) { return st; }
	const vgx_sizes sz = ctx->hostTotals->sizes;
	// Polyline scratch doubles as the heap of the single-pass path (k_flatten_build). Its waves switch to a fresh block
	// when a chunk does not fit: the unused tail of the old block is smaller than that chunk (<= 1x the real vertices
	// over the whole batch), the moved prefix of a spanning sub-path is < VGX_LONG_SUBPATH per >= VGX_BUILD_BLOCK block
	// (<= 1/4), and longer sub-paths grow geometrically (<= 4x their own size). Plus every wave's last open block.
	uint64_t heapVerts = sz.num_poly_vertices * 9 / 4 + 4 * ctx->hostTotals->long_subpath_vertices + 2 * (uint64_t)VGX_BUILD_WAVES * VGX_BUILD_BLOCK;
	// k_flatten_thin (lineTo-only path sets) places draw d at its command prefix -- one slot per command instance -- and the exact
	// builder's draws behind that
	if (ps->thinStatic && heapVerts < 2 * sz.num_cmd_instances + 4096) { heapVerts = 2 * sz.num_cmd_instances + 4096; }
	if (instPeriodFor(ctx, ndraws) || instGroupedFor(ctx, ps, ndraws)) {
		// k_flatten_inst's lane-private blocks (vgx_inst.h): a block left behind wastes less than the one sub-path that did
		// not fit (< the block's useful vertices while sub-paths are at most half a block long), longer sub-paths grow
		// geometrically (<= 4x their size + a block, counted with head room), plus every lane's last open block
		const uint64_t V = sz.num_poly_vertices;
		const uint64_t grown = (ctx->optInstBlock >= VGX_INST_BLOCK) ? V * 9 / 4 + 8 * ctx->hostTotals->inst_long_subpath_vertices : V * 10;
		const uint64_t instVerts = grown + (uint64_t)ctx->optInstWaves * 64 * ctx->optInstBlock + 4096;
		if (instVerts > heapVerts) { heapVerts = instVerts; }
	}
	if ((st = ensureMeshBuffers(ctx, heapVerts, sz.num_subpaths, sz.num_meshes)) != VGX_OK) { return st; }
	// the one-walk route for vgx_tessellate's flatten stage? Unrelated draws (no instancing), curves (not a lineTo-only set), and long ones:
	// >= 10 polyline vertices per command instance (VGX_TESS_FLAT1=2: whatever their length; the two routes cross between 7.5 and 12.6, profiles/experiments/r06_cubics_tessellate_boxes.txt). Sized here, so that the steady state allocates nothing.
	ctx->f1Route = false;
	if (ctx->optTessFlat1 && !ctx->optTwoPass && ndraws > VGX_SMALL_DRAWS && !instPeriodFor(ctx, ndraws) && !instGroupedFor(ctx, ps, ndraws)
		&& !(ps->thinStatic && ctx->optThinStatic) && sz.num_cmd_instances != 0
		&& (ctx->optTessFlat1 == 2 || sz.num_poly_vertices >= 10 * sz.num_cmd_instances)) {
		int cap = 1664; uint32_t segMax = 64; // as vgx_flatten picks them, from the count's own figures
		const double perChunk = (double)sz.num_poly_vertices / (double)sz.num_cmd_instances * 64.0;
		if (perChunk <= 800.0) { cap = 1024; }
		else if (perChunk > 1500.0) {
			cap = 3072;
			const double m = 0.8 * 3072.0 / (perChunk / 64.0);
			segMax = m >= 64.0 ? 64u : (m >= 32.0 ? 32u : (m >= 16.0 ? 16u : 8u));
		}
		if (ctx->optF1Cap) { cap = ctx->optF1Cap; }
		if (ctx->optF1Seg) { segMax = (uint32_t)ctx->optF1Seg; }
		const uint64_t segBound = f1RouteSegmentsFor(ps, ctx->capDraws > ndraws ? ctx->capDraws : ndraws, segMax);
		if (segBound <= (1ull << 24)) {
			if ((st = ensure(ctx, ctx->f1SegDraw, (segBound + 1) * sizeof(uint64_t))) != VGX_OK) { return st; }
			if ((st = ensure(ctx, ctx->f1Segs, (segBound + segBound / 64 + 2) * sizeof(VgxF1Seg))) != VGX_OK) { return st; }
			if ((st = ensure(ctx, ctx->serialList, (ctx->capDraws + 1) * sizeof(uint32_t))) != VGX_OK) { return st; }
			ctx->f1Route = true; ctx->f1RoutePs = ps; ctx->f1RoutePsGen = ps->gen; ctx->f1RouteCap = cap; ctx->f1RouteSegMax = segMax; ctx->f1RouteSegBound = segBound;
		}
	}
	VgxFlattenArgs a = flattenArgs(ctx, ps, draws, ndraws, 1);
	vgx_launch_flatten(true, a, VGX_GRID_BLOCKS, s);
	mark(ctx, s, "flatten_emit");
	VgxCaps outCaps = ctx->caps;
	outCaps.vertices = ~0ull; outCaps.indices = ~0ull;
	runStrokeCount(ctx, draws, outCaps, 0, s);
	if ((st = readTotals(ctx, s)) != VGX_OK) { return st; }
	*out_sizes = ctx->hostTotals->sizes;
	if (ctx->hostTotals->status != VGX_OK) { return (int)ctx->hostTotals->status; } // _emit must not follow a failed count
	ctx->lastPs = ps; ctx->lastDraws = draws; ctx->lastNDraws = ndraws; ctx->lastStage = 2;
	// what the scan over the counted meshes found: a batch with open / Bevel / Round / non-AA strokes is k_fill's and k_stroke's, and the
	// calls that follow a count like that do not launch the tile kernel at all (its 200 000 workgroups would only find that out again)
	ctx->tileHint = ctx->hostTotals->has_general_stroke == 0u;
	return VGX_OK;
}

int vgx_tessellate_emit(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, const vgx_mesh_out* out, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !ps || !out || (!draws && ndraws) || !out->pos || !out->color || !out->idx) {
		return VGX_E_INVALID_ARG;
	}
	if (ctx->lastStage != 2 || ctx->lastPs != ps || ctx->lastDraws != draws || ctx->lastNDraws != ndraws) {
		return VGX_E_INVALID_ARG; // must follow vgx_tessellate_count on the same batch
	}
	const vgx_sizes& sz = ctx->hostTotals->sizes;
	if (out->cap_vertices < sz.num_vertices || out->cap_indices < sz.num_indices || (out->meshes && out->cap_meshes < sz.num_meshes)) {
		return VGX_E_NOSPACE;
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	if (tmplFor(ctx, ps, ndraws)) { return runTmpl(ctx, ps, draws, ndraws, out, nullptr, nullptr, s); }
	return runStrokeEmit(ctx, draws, out, s);
}

int vgx_tessellate(vgx_ctx* ctx, const vgx_pathset* ps, const vgx_draw* draws, uint64_t ndraws, const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !ps || !out || (!draws && ndraws) || !out->pos || !out->color || !out->idx) {
		return VGX_E_INVALID_ARG;
	}
	if (tmplFor(ctx, ps, ndraws)) {
		hipStream_t s = (hipStream_t)stream;
		markBegin(ctx, s);
		ctx->lastStage = 0;
		return runTmpl(ctx, ps, draws, ndraws, out, dev_sizes, dev_status, s);
	}
	if (ndraws > ctx->capDraws || !ctx->cmdCnt.p || !ctx->subFirst.p || !ctx->poly.p || !ctx->mtab.p) {
		return VGX_E_NOSPACE; // scratch was never sized for a batch like this: run vgx_tessellate_count once
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	ctx->lastStage = 0;
	VgxCaps outCaps = ctx->caps;
	outCaps.vertices = out->cap_vertices;
	outCaps.indices = out->cap_indices;
	if (out->meshes && out->cap_meshes < outCaps.meshes) { outCaps.meshes = out->cap_meshes; }
	if (ndraws <= VGX_SMALL_DRAWS && !ctx->optTwoPass && !ctx->optNoSmall) { // (round 6: also with draw-command assembly armed -- runStrokeEmit runs the partition kernels in front of the emit either way)
		// frame-sized batch: five launches instead of seventeen (vgx_flatten.hip, "Frame-sized batches")
		OpCmdPrefix opC;
		opC.draws = draws; opC.pathCmdBegin = ps->dev.path_cmd_begin; opC.npaths = ps->dev.npaths; opC.ndraws = ndraws;
		opC.prefix = (uint64_t*)ctx->cmdPrefix.p; opC.totals = (VgxTotals*)ctx->totals.p; opC.cap = ctx->caps.cmd_instances;
		opC.period = 0; opC.pathSubBegin = ps->dev.path_sub_begin; opC.subPrefix = (uint64_t*)ctx->subPrefix.p;
		vgx_launch_small_front(&opC, (vgx_draw_info*)ctx->dinfo.p, s);
		mark(ctx, s, "small_front");
		VgxFlattenArgs f = flattenArgs(ctx, ps, draws, ndraws, 1);
		f.build_mode = 1;
		f.mprep = (VgxMeshPrep*)ctx->mprep.p;
		vgx_launch_flatten_build(f, ctx->optBuildWaves, s, false);
		mark(ctx, s, "flatten_build");
		VgxStrokeArgs sa;
		sa.draws = draws; sa.poly = (const float*)ctx->poly.p; sa.mdesc = (const VgxMeshDesc*)ctx->mdesc.p;
		sa.elem_prefix = nullptr; sa.elem_prefix_fill = (const uint64_t*)ctx->elemPrefix.p; sa.elem_prefix_stroke = (const uint64_t*)ctx->elemPrefixS.p;
		sa.mprep = (VgxMeshPrep*)ctx->mprep.p; sa.mtab = (vgx_mesh*)ctx->mtab.p;
		sa.pos = nullptr; sa.color = nullptr; sa.idx = nullptr; sa.meshes_out = nullptr; sa.mesh_base = nullptr;
		sa.totals = (VgxTotals*)ctx->totals.p; sa.caps = outCaps; sa.tile_mode = 0; sa.no_long = 1;
		OpDrawInfo opD;
		opD.dinfo = (vgx_draw_info*)ctx->dinfo.p; opD.ndraws = ndraws; opD.totals = (VgxTotals*)ctx->totals.p; opD.caps = ctx->caps; opD.keepPolyBase = 1;
		OpMeshAll opM;
		opM.mdesc = (const VgxMeshDesc*)ctx->mdesc.p; opM.mtab = (vgx_mesh*)ctx->mtab.p; opM.meshesOut = out->meshes;
		opM.prefixFill = (uint64_t*)ctx->elemPrefix.p; opM.prefixStroke = (uint64_t*)ctx->elemPrefixS.p;
		opM.totals = (VgxTotals*)ctx->totals.p; opM.caps = outCaps; opM.checkCaps = 1; opM.fixedSize = 0; opM.fixedCount = 0;
		vgx_launch_small_middle(f, sa, &opD, &opM, dev_sizes, dev_status, s);
		mark(ctx, s, "small_middle");
		const int est = runStrokeEmit(ctx, draws, out, s, nullptr, true);
		if (est != VGX_OK) { return est; }
		if (ctx->asmArmed && (dev_sizes || dev_status)) { // the partition kernels may have ended the call (a mesh beyond any vertex buffer, the command table full): the verdict once more
			hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, s, (const VgxTotals*)ctx->totals.p, dev_sizes, dev_status);
		}
		return launchStatus(ctx);
	}
	const bool oneWalk = ctx->f1Route && ps == ctx->f1RoutePs && ps->gen == ctx->f1RoutePsGen && !ctx->optTwoPass && !instPeriodFor(ctx, ndraws) && !instGroupedFor(ctx, ps, ndraws)
		&& f1RouteSegmentsFor(ps, ndraws, ctx->f1RouteSegMax) <= ctx->f1RouteSegBound; // (what the last count decided and sized, for this path set)
	if (!oneWalk) { runCmdPrefix(ctx, ps, draws, ndraws, s, ctx->optTwoPass ? 0u : instPeriodFor(ctx, ndraws)); }
	if (oneWalk) {
		runFlattenOneWalk(ctx, ps, draws, ndraws, s);
	} else if (ctx->optTwoPass) { // tuning / debugging knob: the ordered two-pass flatten
		runFlattenCount(ctx, ps, draws, ndraws, s);
		VgxFlattenArgs a = flattenArgs(ctx, ps, draws, ndraws, 1);
		vgx_launch_flatten(true, a, VGX_GRID_BLOCKS, s);
		mark(ctx, s, "flatten_emit");
	} else {
		runFlattenBuild(ctx, ps, draws, ndraws, s);
	}
	runStrokeCount(ctx, draws, outCaps, 1, s, nullptr, !ctx->optTwoPass, out->meshes);
	{
		const int st = runStrokeEmit(ctx, draws, out, s, nullptr, true);
		if (st != VGX_OK) { return st; }
	}
	if (dev_sizes || dev_status) {
		hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, s, (const VgxTotals*)ctx->totals.p, dev_sizes, dev_status);
	}
	return launchStatus(ctx);
}

// ---- stroker-level entry ------------------------------------------------------------------------------
int vgx_stroke_count(vgx_ctx* ctx, const float* poly, const vgx_subpath* subpaths, const uint32_t* subpath_draw, uint64_t nsubpaths, const vgx_draw* draws, uint64_t ndraws, vgx_sizes* out_sizes, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !out_sizes || (nsubpaths && (!poly || !subpaths || !subpath_draw || !draws))) {
		return VGX_E_INVALID_ARG;
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	ctx->lastStage = 0;
	int st;
	if ((st = ensure(ctx, ctx->partial, VGX_SCAN_BLOCKS * sizeof(Sum3))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->totals, sizeof(VgxTotals))) != VGX_OK) { return st; }
	// at most two meshes per vertex list: scratch can be sized without a device round trip
	if ((st = ensureMeshBuffers(ctx, 0, 0, 2 * nsubpaths)) != VGX_OK) { return st; }
	noteHip(ctx, hipMemsetAsync(ctx->totals.p, 0, sizeof(VgxTotals), s));
	OpSubMeshes op;
	op.subs = subpaths; op.subDraw = subpath_draw; op.draws = draws; op.nsubs = nsubpaths; op.ndraws = ndraws;
	op.mdesc = (VgxMeshDesc*)ctx->mdesc.p; op.mtab = (vgx_mesh*)ctx->mtab.p; op.totals = (VgxTotals*)ctx->totals.p;
	vgx_device_scan(op, (Sum3*)ctx->partial.p, s, nsubpaths);
	mark(ctx, s, "scan_subpath_meshes");
	VgxCaps outCaps = ctx->caps;
	outCaps.vertices = ~0ull; outCaps.indices = ~0ull;
	runStrokeCount(ctx, draws, outCaps, 0, s, poly);
	if ((st = readTotals(ctx, s)) != VGX_OK) { return st; }
	*out_sizes = ctx->hostTotals->sizes;
	if (ctx->hostTotals->status != VGX_OK) { return (int)ctx->hostTotals->status; }
	ctx->lastPs = nullptr; ctx->lastDraws = draws; ctx->lastNDraws = nsubpaths; ctx->lastStage = 3;
	return VGX_OK;
}

int vgx_stroke_emit(vgx_ctx* ctx, const float* poly, const vgx_subpath* subpaths, const uint32_t* subpath_draw, uint64_t nsubpaths, const vgx_draw* draws, uint64_t ndraws, const vgx_mesh_out* out, void* stream)
{
	DeviceGuard guard(ctx);
	(void)subpaths; (void)subpath_draw; (void)ndraws;
	if (!ctx || !out || !out->pos || !out->color || !out->idx) {
		return VGX_E_INVALID_ARG;
	}
	if (ctx->lastStage != 3 || ctx->lastDraws != draws || ctx->lastNDraws != nsubpaths) {
		return VGX_E_INVALID_ARG; // must follow vgx_stroke_count on the same batch
	}
	const vgx_sizes& sz = ctx->hostTotals->sizes;
	if (out->cap_vertices < sz.num_vertices || out->cap_indices < sz.num_indices || (out->meshes && out->cap_meshes < sz.num_meshes)) {
		return VGX_E_NOSPACE;
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	return runStrokeEmit(ctx, draws, out, s, poly);
}

// ---- shape cache ------------------------------------------------------------------------------------
int vgx_cache_localize(vgx_ctx* ctx, const vgx_draw* draws, uint64_t ndraws, float* pos, const vgx_mesh* meshes, uint64_t num_meshes, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || (num_meshes && (!draws || !pos || !meshes))) {
		return VGX_E_INVALID_ARG;
	}
	if (num_meshes) {
		vgx_launch_cache_localize(draws, ndraws, pos, meshes, num_meshes, (hipStream_t)stream);
	}
	return VGX_OK;
}

int vgx_cache_submit(vgx_ctx* ctx, const vgx_cache_desc* cache, const vgx_cache_instance* instances, uint64_t ninst, const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !cache || !out || (!instances && ninst) || !out->pos || !out->color || !out->idx
		|| (cache->num_meshes && (!cache->pos || !cache->color || !cache->idx || !cache->meshes))) {
		return VGX_E_INVALID_ARG;
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	ctx->lastStage = 0;
	int st;
	if ((st = ensure(ctx, ctx->totals, sizeof(VgxTotals))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->partial, VGX_SCAN_BLOCKS * sizeof(Sum3))) != VGX_OK) { return st; }
	// the three instance prefix arrays share one scratch buffer; the output mesh table lives in the mesh-table scratch
	if ((st = ensure(ctx, ctx->cmdPrefix, 3 * (ninst + 1) * sizeof(uint64_t))) != VGX_OK) { return st; }
	// the internal mesh table must hold every output mesh even when the caller does not want the table
	uint64_t meshCap = out->meshes ? out->cap_meshes : cache->num_meshes * ninst;
	if (meshCap == 0) { meshCap = 1; }
	if ((st = ensure(ctx, ctx->mtab, (meshCap + 1) * sizeof(vgx_mesh))) != VGX_OK) { return st; }
	noteHip(ctx, hipMemsetAsync(ctx->totals.p, 0, sizeof(VgxTotals), s));
	uint64_t* prefix = (uint64_t*)ctx->cmdPrefix.p;
	OpCacheInst op;
	op.cache = *cache; op.inst = instances; op.ninst = ninst;
	op.meshPrefix = prefix; op.vertPrefix = prefix + (ninst + 1); op.idxPrefix = prefix + 2 * (ninst + 1);
	op.totals = (VgxTotals*)ctx->totals.p;
	op.caps = ctx->caps;
	op.caps.meshes = ctx->mtab.cap / sizeof(vgx_mesh) - 1;
	if (out->meshes && out->cap_meshes < op.caps.meshes) { op.caps.meshes = out->cap_meshes; }
	op.caps.vertices = out->cap_vertices; op.caps.indices = out->cap_indices;
	vgx_device_scan(op, (Sum3*)ctx->partial.p, s, ninst);
	mark(ctx, s, "scan_instances");
	VgxCacheArgs a;
	a.cache = *cache; a.inst = instances; a.ninst = ninst;
	a.inst_mesh_prefix = op.meshPrefix; a.inst_vert_prefix = op.vertPrefix; a.inst_idx_prefix = op.idxPrefix;
	a.mtab = (vgx_mesh*)ctx->mtab.p; a.meshes_out = out->meshes;
	a.pos = out->pos; a.color = out->color; a.idx = out->idx;
	a.mesh_base = nullptr;
	a.totals = (VgxTotals*)ctx->totals.p;
	vgx_launch_cache_meshes(a, s);
	mark(ctx, s, "cache_meshes");
	if (ctx->asmArmed) {
		if ((st = runAssemble(ctx, out, s)) != VGX_OK) { return st; }
		a.mesh_base = (const uint32_t*)ctx->meshBase.p;
	}
	vgx_launch_cache_copy(a, vgxElementGrid(out->cap_vertices), s);
	mark(ctx, s, "cache_copy");
	if (dev_sizes || dev_status) {
		hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, s, (const VgxTotals*)ctx->totals.p, dev_sizes, dev_status);
	}
	return launchStatus(ctx);
}

int vgx_merge(vgx_ctx* ctx, const vgx_cache_desc* a, const vgx_cache_desc* b, const uint32_t* b_draw, const vgx_draw* draws, uint64_t ndraws,
              const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream)
{
	return vgx_merge_uv(ctx, a, b, b_draw, nullptr, draws, ndraws, out, dev_sizes, dev_status, stream);
}

int vgx_merge_uv(vgx_ctx* ctx, const vgx_cache_desc* a, const vgx_cache_desc* b, const uint32_t* b_draw, const void* b_uv, const vgx_draw* draws, uint64_t ndraws,
                 const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream)
{
	DeviceGuard guard(ctx);
	(void)ndraws;
	if (!ctx || !a || !b || !out || !out->pos || !out->color || !out->idx
		|| (a->num_meshes && (!a->pos || !a->color || !a->idx || !a->meshes)) || (b->num_meshes && (!b->pos || !b->color || !b->idx || !b->meshes))) {
		return VGX_E_INVALID_ARG;
	}
	const uint64_t n = a->num_meshes + b->num_meshes;
	if (n >= 0x7FFFFFFFull) { return VGX_E_RANGE; }
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	ctx->lastStage = 0;
	ctx->tmplOn = false; // the mesh-table scratch is re-sized below
	int st;
	if ((st = ensure(ctx, ctx->totals, sizeof(VgxTotals))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->partial, VGX_SCAN_BLOCKS * sizeof(Sum3))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->mtab, (n + 1) * sizeof(vgx_mesh))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->mdesc, (n + 1) * sizeof(VgxMeshDesc))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->cmdPrefix, (n + 1) * sizeof(uint64_t))) != VGX_OK) { return st; } // the merged order lives in the command-prefix scratch
	noteHip(ctx, hipMemsetAsync(ctx->totals.p, 0, sizeof(VgxTotals), s));
	VgxMergeArgs m;
	memset(&m, 0, sizeof(m));
	m.a = *a; m.b = *b; m.b_draw = b_draw;
	m.order = (uint32_t*)ctx->cmdPrefix.p; m.mtab = (vgx_mesh*)ctx->mtab.p; m.mdesc = (VgxMeshDesc*)ctx->mdesc.p;
	m.meshes_out = out->meshes; m.pos = out->pos; m.color = out->color; m.idx = out->idx; m.mesh_base = nullptr;
	m.totals = (VgxTotals*)ctx->totals.p;
	m.caps = ctx->caps; m.caps.vertices = out->cap_vertices; m.caps.indices = out->cap_indices; m.caps.meshes = out->cap_meshes;
	vgx_launch_merge_rank(m, s);
	vgx_launch_merge_scan(m, ctx->partial.p, s);
	mark(ctx, s, "merge_scan");
	if (ctx->asmArmed) {
		if ((st = runAssemble(ctx, out, s, draws)) != VGX_OK) { return st; }
		m.mesh_base = (const uint32_t*)ctx->meshBase.p;
		if (b_uv && ctx->asmCfg.uv && ctx->asmCfg.uv_bytes) { m.b_uv = b_uv; m.uv_out = ctx->asmCfg.uv; m.uv_bytes = ctx->asmCfg.uv_bytes; }
	}
	vgx_launch_merge_copy(m, s);
	mark(ctx, s, "merge_copy");
	if (dev_sizes || dev_status) {
		hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, s, (const VgxTotals*)ctx->totals.p, dev_sizes, dev_status);
	}
	return launchStatus(ctx);
}

int vgx_set_assembly(vgx_ctx* ctx, const vgx_assembly* asm_)
{
	if (!ctx) {
		return VGX_E_INVALID_ARG;
	}
	if (!asm_) {
		ctx->asmArmed = false;
		return VGX_OK;
	}
	if (!asm_->drawcmds || asm_->cap_drawcmds == 0 || asm_->max_vb_vertices > 65536u) { // vg.cpp:734: indices are uint16
		return VGX_E_INVALID_ARG;
	}
	if ((asm_->flags & ~(uint32_t)VGX_ASM_SPLIT_STATE) || (asm_->uv && asm_->uv_bytes != 4 && asm_->uv_bytes != 8)) {
		return VGX_E_INVALID_ARG;
	}
	ctx->asmCfg = *asm_;
	ctx->asmArmed = true;
	return VGX_OK;
}

// ---- concave fills ------------------------------------------------------------------------------------
int vgx_concave_move(vgx_ctx* ctx, const float* contour_verts, uint64_t num_contour_vertices, const vgx_contour* contours, uint64_t ncontours,
                     const vgx_concave_fill* fills, uint64_t nfills, float* moved, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || (num_contour_vertices && (!contour_verts || !contours || !fills || !moved || !ncontours || !nfills))) {
		return VGX_E_INVALID_ARG;
	}
	if (!num_contour_vertices) { return VGX_OK; }
	VgxConcaveArgs a;
	memset(&a, 0, sizeof(a));
	a.contour_verts = contour_verts; a.contours = contours; a.ncontours = ncontours; a.num_contour_vertices = num_contour_vertices;
	a.fills = fills; a.nfills = nfills; a.moved = moved;
	vgx_launch_concave_move(a, (hipStream_t)stream);
	return launchStatus(ctx);
}

int vgx_concave_emit(vgx_ctx* ctx, const float* contour_verts, uint64_t num_contour_vertices, const vgx_contour* contours, uint64_t ncontours,
                     const vgx_concave_fill* fills, uint64_t nfills, const float* tess_pos, const uint16_t* tess_idx,
                     const vgx_mesh_out* out, vgx_sizes* dev_sizes, uint32_t* dev_status, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !out || !out->pos || !out->color || !out->idx || (nfills && !fills) || (ncontours && (!contours || !contour_verts))) {
		return VGX_E_INVALID_ARG;
	}
	hipStream_t s = (hipStream_t)stream;
	markBegin(ctx, s);
	ctx->lastStage = 0;
	int st;
	if ((st = ensure(ctx, ctx->totals, sizeof(VgxTotals))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->partial, VGX_SCAN_BLOCKS * sizeof(Sum3))) != VGX_OK) { return st; }
	if ((st = ensure(ctx, ctx->mtab, (nfills + 1) * sizeof(vgx_mesh))) != VGX_OK) { return st; }
	noteHip(ctx, hipMemsetAsync(ctx->totals.p, 0, sizeof(VgxTotals), s));
	OpConcaveFills op;
	op.contours = contours; op.ncontours = ncontours; op.fills = fills; op.nfills = nfills;
	op.mtab = (vgx_mesh*)ctx->mtab.p; op.meshesOut = out->meshes; op.totals = (VgxTotals*)ctx->totals.p;
	op.caps = ctx->caps;
	op.caps.vertices = out->cap_vertices; op.caps.indices = out->cap_indices; op.caps.meshes = out->meshes ? out->cap_meshes : ~0ull;
	vgx_device_scan(op, (Sum3*)ctx->partial.p, s, nfills);
	mark(ctx, s, "scan_concave_fills");
	VgxConcaveArgs a;
	memset(&a, 0, sizeof(a));
	a.contour_verts = contour_verts; a.contours = contours; a.ncontours = ncontours; a.num_contour_vertices = num_contour_vertices;
	a.fills = fills; a.nfills = nfills; a.tess_pos = tess_pos; a.tess_idx = tess_idx;
	a.mtab = (const vgx_mesh*)ctx->mtab.p; a.pos = out->pos; a.color = out->color; a.idx = out->idx;
	a.totals = (VgxTotals*)ctx->totals.p;
	vgx_launch_concave_emit(a, s);
	mark(ctx, s, "concave_emit");
	if (dev_sizes || dev_status) {
		hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, s, (const VgxTotals*)ctx->totals.p, dev_sizes, dev_status);
	}
	return launchStatus(ctx);
}

int vgx_get_failure_info(vgx_ctx* ctx, vgx_failure_info* out, void* stream)
{
	if (!ctx || !out) {
		return VGX_E_INVALID_ARG;
	}
	DeviceGuard guard(ctx);
	memset(out, 0, sizeof(*out));
	if (!ctx->totals.p) {
		return VGX_OK;
	}
	const int st = readTotals(ctx, (hipStream_t)stream);
	if (st != VGX_OK) { return st; }
	out->status = ctx->hostTotals->status;
	out->reason = ctx->hostTotals->fail_reason;
	out->aux = ctx->hostTotals->fail_aux;
	out->segment = ctx->hostTotals->fail_segment;
	out->segment_items = ctx->tmplOn ? 5u : ctx->optInst ? (ctx->instPeriod ? (ctx->instPermOn ? 4u : 1u) : (ctx->instGrouped ? (ctx->instClasses > 1 ? 3u : 2u) : 0u)) : 0u; // flatten mode chosen by the last count call
	if (out->segment_items == 0u && ctx->lastPs && ctx->lastPs->thinStatic && ctx->optThinStatic) { out->segment_items = 6u; } // k_flatten_thin in k_flatten_build's place
	for (int i = 0; i < 16; ++i) { out->prof[i] = ctx->hostTotals->prof[i]; }
	return VGX_OK;
}

int vgx_set_static_batches(vgx_ctx* ctx, int enable)
{
	if (!ctx) {
		return VGX_E_INVALID_ARG;
	}
	ctx->optTmplBatch = enable ? 1 : 0;
	if (!enable && ctx->tmplIsBatch) { ctx->tmplOn = false; } // the next vgx_tessellate wants a count again
	return VGX_OK;
}

int vgx_set_profiling(vgx_ctx* ctx, int enable)
{
	if (!ctx) {
		return VGX_E_INVALID_ARG;
	}
	ctx->profiling = enable ? 1 : 0;
	ctx->profCalls = 0;
	return VGX_OK;
}

int vgx_get_stage_times(vgx_ctx* ctx, vgx_stage_times* out)
{
	return vgx_get_stage_times_avg(ctx, out, 1);
}

int vgx_get_stage_times_avg(vgx_ctx* ctx, vgx_stage_times* out, uint32_t ncalls)
{
	DeviceGuard guard(ctx);
	if (!ctx || !out) {
		return VGX_E_INVALID_ARG;
	}
	memset(out, 0, sizeof(*out));
	if (!ctx->profiling || !ctx->evCreated || ctx->profCalls == 0) {
		return VGX_OK;
	}
	uint64_t n = ncalls < 1 ? 1 : ncalls;
	if (n > VGX_PROF_RING) { n = VGX_PROF_RING; }
	if (n > ctx->profCalls) { n = ctx->profCalls; }
	const uint32_t last = (uint32_t)((ctx->profCalls - 1) % VGX_PROF_RING);
	out->num_stages = ctx->numEv[last];
	uint32_t used = 0;
	for (uint64_t c = 0; c < n; ++c) {
		const uint32_t k = (uint32_t)((ctx->profCalls - 1 - c) % VGX_PROF_RING);
		if (ctx->numEv[k] != ctx->numEv[last]) { continue; } // a call of another kind in between: not averaged in
		bool same = true;
		for (uint32_t i = 0; i < ctx->numEv[k]; ++i) { same = same && ctx->evName[k][i] == ctx->evName[last][i]; }
		if (!same) { continue; }
		for (uint32_t i = 0; i < ctx->numEv[k]; ++i) {
			float ms = 0.0f;
			if (hipEventElapsedTime(&ms, ctx->ev[k][i], ctx->ev[k][i + 1]) != hipSuccess) { ms = -1.0f; }
			out->ms[i] += ms;
		}
		++used;
	}
	for (uint32_t i = 0; i < out->num_stages; ++i) {
		out->ms[i] = used ? out->ms[i] / (float)used : -1.0f;
		out->name[i] = ctx->evName[last][i];
	}
	return VGX_OK;
}

} // extern "C"

// ---- multi-GPU gather over RCCL (the layout of vg-renderer_amd/dist.py behind the C-ABI) ------------------
// librccl is bound with dlopen / dlsym from the copy the process already has (a C++ host links it; a torch process
// carries its own), never linked: libvgx.so loads on boxes without it and cannot pull a second copy into a process.
#include <dlfcn.h>

struct VgxRccl
{
	void* lib;
	int (*GroupStart)();
	int (*GroupEnd)();
	int (*Send)(const void*, size_t, int, int, void*, hipStream_t);
	int (*Recv)(void*, size_t, int, int, void*, hipStream_t);
	int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t);
	int (*CommCount)(void*, int*);
	int (*CommUserRank)(void*, int*);
};
enum { kNcclUint8 = 1, kNcclUint64 = 5 }; // ncclDataType_t values (rccl.h)

static void vgx_rccl_release(vgx_ctx* ctx)
{
	if (ctx->rccl) {
		if (ctx->rccl->lib) { (void)dlclose(ctx->rccl->lib); }
		delete ctx->rccl;
		ctx->rccl = nullptr;
	}
}

namespace {

int bindRccl(vgx_ctx* ctx)
{
	if (ctx->rccl) {
		return VGX_OK;
	}
	void* h = nullptr;
	if (const char* over = getenv("VGX_RCCL_LIB")) { h = dlopen(over, RTLD_NOW | RTLD_LOCAL); } // testing knob: tests/native/fake_rccl.cpp
	const char* names[] = { "librccl.so.1", "librccl.so" };
	for (const char* n : names) { if (!h) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); } } // the copy the process already uses
	for (const char* n : names) { if (!h) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); } }
	if (!h) { h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL); }
	if (!h) {
		return VGX_E_NO_DEVICE;
	}
	VgxRccl* r = new VgxRccl;
	r->lib = h;
	r->GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
	r->GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
	r->Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
	r->Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
	r->AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
	r->CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
	r->CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
	if (!r->GroupStart || !r->GroupEnd || !r->Send || !r->Recv || !r->AllGather || !r->CommCount || !r->CommUserRank) {
		(void)dlclose(h);
		delete r;
		return VGX_E_NO_DEVICE;
	}
	ctx->rccl = r;
	return VGX_OK;
}

#define RCCLCHK(ctx, call)                                    \
	do {                                                      \
		const int r_ = (call);                                \
		if (r_ != 0) {                                        \
			(ctx)->lastHipError = 10000 + r_;                 \
			return VGX_E_HIP;                                 \
		}                                                     \
	} while (0)

// mesh records of rank r sit at [mesh0, mesh1): add the rank's vertex / index / draw bases
struct GatherRebase { uint64_t mesh0, mesh1, vbase, ibase; uint32_t dbase; };
#define VGX_GATHER_MAX_RANKS 64
struct GatherRebaseArgs { vgx_mesh* meshes; int n; GatherRebase r[VGX_GATHER_MAX_RANKS]; };

__global__ __launch_bounds__(256) void k_gather_rebase(GatherRebaseArgs A)
{
	const GatherRebase R = A.r[blockIdx.y];
	for (uint64_t m = R.mesh0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m < R.mesh1; m += (uint64_t)gridDim.x * blockDim.x) {
		vgx_mesh rec = A.meshes[m];
		rec.first_vertex += R.vbase;
		rec.first_index += R.ibase;
		rec.draw += R.dbase;
		A.meshes[m] = rec;
	}
}

} // namespace

// Transfers larger than this leave in several pieces: piece c of every (rank, stream) pair goes out in group c, so that no
// single RCCL operation is gigabytes long and the proxy threads can pipeline (VGX_GATHER_CHUNK_MB, default 256 MiB).
static size_t gatherChunkBytes()
{
	static size_t v = 0;
	if (!v) {
		const char* e = getenv("VGX_GATHER_CHUNK_MB");
		const long mb = e ? atol(e) : 256;
		v = (size_t)(mb >= 1 ? mb : 256) << 20;
	}
	return v;
}

extern "C" int vgx_gather_sizes(vgx_ctx* ctx, void* rccl_comm, const vgx_rank_sizes* mine, vgx_rank_sizes* all, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !rccl_comm || !mine || !all) {
		return VGX_E_INVALID_ARG;
	}
	int st = bindRccl(ctx);
	if (st != VGX_OK) { return st; }
	int nranks = 0, rank = 0;
	RCCLCHK(ctx, ctx->rccl->CommCount(rccl_comm, &nranks));
	RCCLCHK(ctx, ctx->rccl->CommUserRank(rccl_comm, &rank));
	if (nranks < 1 || rank < 0 || rank >= nranks) { return VGX_E_INVALID_ARG; }
	// five words per rank: the four totals + the rank's transfer piece size. The piece size is part of the wire protocol of
	// vgx_gather_at (sender and root must cut a stream into the same pieces) but comes from each process's own environment
	// (VGX_GATHER_CHUNK_MB): ranks that disagree are told here, before a gather could hang on mismatched Send / Recv sizes.
	struct Wire { vgx_rank_sizes z; uint64_t chunk; };
	if ((st = ensure(ctx, ctx->gatherSizes, (size_t)nranks * sizeof(Wire))) != VGX_OK) { return st; }
	hipStream_t s = (hipStream_t)stream;
	Wire* dev = (Wire*)ctx->gatherSizes.p;
	std::vector<Wire> host((size_t)nranks);
	host[(size_t)rank].z = *mine; host[(size_t)rank].chunk = gatherChunkBytes();
	HIPCHK(ctx, hipMemcpyAsync(dev + rank, &host[(size_t)rank], sizeof(Wire), hipMemcpyHostToDevice, s));
	RCCLCHK(ctx, ctx->rccl->AllGather(dev + rank, dev, 5, kNcclUint64, rccl_comm, s)); // in place: my slot is my send buffer
	HIPCHK(ctx, hipMemcpyAsync(host.data(), dev, (size_t)nranks * sizeof(Wire), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	for (int r = 0; r < nranks; ++r) {
		all[r] = host[(size_t)r].z;
		if (host[(size_t)r].chunk != gatherChunkBytes()) { return VGX_E_INVALID_ARG; }
	}
	return VGX_OK;
}

extern "C" int vgx_gather_at(vgx_ctx* ctx, void* rccl_comm, int root, const vgx_mesh_out* local, const vgx_rank_sizes* all, const vgx_rank_sizes* place,
	const vgx_mesh_out* global, void* stream)
{
	DeviceGuard guard(ctx);
	if (!ctx || !rccl_comm || !local || !all || !place) {
		return VGX_E_INVALID_ARG;
	}
	int st = bindRccl(ctx);
	if (st != VGX_OK) { return st; }
	int nranks = 0, rank = 0;
	RCCLCHK(ctx, ctx->rccl->CommCount(rccl_comm, &nranks));
	RCCLCHK(ctx, ctx->rccl->CommUserRank(rccl_comm, &rank));
	if (nranks < 1 || nranks > VGX_GATHER_MAX_RANKS || root < 0 || root >= nranks) { return VGX_E_INVALID_ARG; }
	hipStream_t s = (hipStream_t)stream;
	const vgx_rank_sizes& me = all[rank];
	if (me.num_vertices > local->cap_vertices || me.num_indices > local->cap_indices || (local->meshes && me.num_meshes > local->cap_meshes)) {
		return VGX_E_INVALID_ARG; // the sizes do not describe `local`
	}
	if ((me.num_vertices && (!local->pos || !local->color)) || (me.num_indices && !local->idx) || (me.num_meshes && !local->meshes)) {
		return VGX_E_INVALID_ARG;
	}
	if (rank == root) {
		if (!global || !global->pos || !global->color || !global->idx || !global->meshes) { return VGX_E_INVALID_ARG; }
		for (int r = 0; r < nranks; ++r) { // every block inside the destination
			if (place[r].num_vertices + all[r].num_vertices > global->cap_vertices || place[r].num_indices + all[r].num_indices > global->cap_indices
				|| place[r].num_meshes + all[r].num_meshes > global->cap_meshes) {
				return VGX_E_NOSPACE;
			}
		}
	}
	// the four streams of one rank's block as byte ranges: {local source, destination in global, bytes}
	struct Piece { const uint8_t* src; uint8_t* dst; size_t bytes; };
	auto blockOf = [&](int r, Piece out[4]) {
		const vgx_rank_sizes& z = all[r];
		const vgx_rank_sizes& o = place[r];
		const bool isRoot = rank == root;
		out[0] = Piece{ (const uint8_t*)local->pos, isRoot ? (uint8_t*)(global->pos + 2 * o.num_vertices) : nullptr, (size_t)z.num_vertices * 8 };
		out[1] = Piece{ (const uint8_t*)local->color, isRoot ? (uint8_t*)(global->color + o.num_vertices) : nullptr, (size_t)z.num_vertices * 4 };
		out[2] = Piece{ (const uint8_t*)local->idx, isRoot ? (uint8_t*)(global->idx + o.num_indices) : nullptr, (size_t)z.num_indices * 2 };
		out[3] = Piece{ (const uint8_t*)local->meshes, isRoot ? (uint8_t*)(global->meshes + o.num_meshes) : nullptr, (size_t)z.num_meshes * sizeof(vgx_mesh) };
	};
	const size_t chunk = gatherChunkBytes();
	size_t maxBytes = 0;
	for (int r = 0; r < nranks; ++r) {
		if (rank != root && r != rank) { continue; }
		Piece b[4]; blockOf(r, b);
		for (int k = 0; k < 4; ++k) { if (b[k].bytes > maxBytes) { maxBytes = b[k].bytes; } }
	}
	const size_t nchunks = maxBytes ? (maxBytes + chunk - 1) / chunk : 0;
	// Inside a group a failed call must not return at once: the group stays open on this thread and every later collective
	// would be queued and never issued. Remember the first error, skip the rest, always close the group.
	int rcclErr = 0;
#define RCCLGRP(call) do { if (!rcclErr) { rcclErr = (call); } } while (0)
	if (rank == root) { // my own block: plain copies on the same stream (they overlap with the incoming transfers)
		Piece b[4]; blockOf(root, b);
		for (int k = 0; k < 4; ++k) { if (b[k].bytes) { noteHip(ctx, hipMemcpyAsync(b[k].dst, b[k].src, b[k].bytes, hipMemcpyDeviceToDevice, s)); } }
	}
	for (size_t c = 0; c < nchunks && !rcclErr; ++c) {
		RCCLCHK(ctx, ctx->rccl->GroupStart());
		for (int r = 0; r < nranks; ++r) {
			if (r == root || (rank != root && r != rank)) { continue; }
			Piece b[4]; blockOf(r, b);
			for (int k = 0; k < 4; ++k) {
				const size_t off = c * chunk;
				if (off >= b[k].bytes) { continue; }
				const size_t n = b[k].bytes - off < chunk ? b[k].bytes - off : chunk;
				if (rank == root) { RCCLGRP(ctx->rccl->Recv(b[k].dst + off, n, kNcclUint8, r, rccl_comm, s)); }
				else { RCCLGRP(ctx->rccl->Send(b[k].src + off, n, kNcclUint8, root, rccl_comm, s)); }
			}
		}
		const int endErr = ctx->rccl->GroupEnd();
		if (!rcclErr) { rcclErr = endErr; }
	}
#undef RCCLGRP
	if (rcclErr) { ctx->lastHipError = 10000 + rcclErr; return VGX_E_HIP; }
	if (rank == root) { // mesh records of rank r sit at [place, place + n): add the block's vertex / index / draw bases
		GatherRebaseArgs ra;
		ra.meshes = global->meshes;
		ra.n = 0;
		uint64_t maxMeshes = 0;
		for (int r = 0; r < nranks; ++r) {
			const vgx_rank_sizes& z = all[r];
			const vgx_rank_sizes& o = place[r];
			if (z.num_meshes && (o.num_vertices || o.num_indices || o.num_draws)) {
				GatherRebase& g = ra.r[ra.n++];
				g.mesh0 = o.num_meshes; g.mesh1 = o.num_meshes + z.num_meshes; g.vbase = o.num_vertices; g.ibase = o.num_indices; g.dbase = (uint32_t)o.num_draws;
				if (z.num_meshes > maxMeshes) { maxMeshes = z.num_meshes; }
			}
		}
		if (ra.n) {
			const uint64_t blocks = (maxMeshes + 255) / 256;
			hipLaunchKernelGGL(k_gather_rebase, dim3((unsigned)(blocks > 4096 ? 4096 : blocks), (unsigned)ra.n), dim3(256), 0, s, ra);
		}
	}
	if (ctx->pendingHipError) { ctx->lastHipError = ctx->pendingHipError; ctx->pendingHipError = 0; return VGX_E_HIP; }
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) { ctx->lastHipError = (int)e; return VGX_E_HIP; }
	return VGX_OK;
}

extern "C" int vgx_gather(vgx_ctx* ctx, void* rccl_comm, int root, const vgx_mesh_out* local, const vgx_rank_sizes* all, const vgx_mesh_out* global, void* stream)
{
	if (!ctx || !rccl_comm || !local || !all) {
		return VGX_E_INVALID_ARG;
	}
	int st;
	{
		DeviceGuard guard(ctx);
		st = bindRccl(ctx);
	}
	if (st != VGX_OK) { return st; }
	int nranks = 0;
	RCCLCHK(ctx, ctx->rccl->CommCount(rccl_comm, &nranks));
	if (nranks < 1 || nranks > VGX_GATHER_MAX_RANKS) { return VGX_E_INVALID_ARG; }
	vgx_rank_sizes place[VGX_GATHER_MAX_RANKS];
	uint64_t vo = 0, io = 0, mo = 0, dofs = 0;
	for (int r = 0; r < nranks; ++r) { // rank order = draw order: exclusive prefix of the sizes
		place[r].num_vertices = vo; place[r].num_indices = io; place[r].num_meshes = mo; place[r].num_draws = dofs;
		vo += all[r].num_vertices; io += all[r].num_indices; mo += all[r].num_meshes; dofs += all[r].num_draws;
	}
	return vgx_gather_at(ctx, rccl_comm, root, local, all, place, global, stream);
}
