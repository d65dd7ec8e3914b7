// gather_test.cpp -- vgx_gather_sizes / vgx_gather driven from C++ the way a multi-GPU host would (one context + one
// communicator per rank), checked against ONE context tessellating the whole batch: the gathered streams must be
// byte-identical and the gathered mesh table must equal the single-context table (first_vertex / first_index / draw
// rebased by the ranks in front).
//   gather_test real            1 rank, the real librccl (ncclCommInitRank with nranks = 1): binding, collective, copies
//   gather_test fake LIB N [D]  N ranks as threads of this process on one GPU over tests/native/fake_rccl.cpp (VGX_RCCL_LIB=LIB), D draws
//   gather_test tiles LIB N D T the same with every rank's draws cut into T tiles: tile t is tessellated into the local buffers
//                               behind tile t - 1 and gathered by vgx_gather_at (place = rank base + tiles in front), small
//                               VGX_GATHER_CHUNK_MB so that every transfer leaves in several pieces
//   gather_test big LIB N GB    N ranks gather the SAME local block of about GB gigabytes each into one destination (8 x 8.75 GB =
//                               the Tiger x80k root of BASELINE config 4): 64-bit offsets, capacity checks, piece splitting at size
// Build: hipcc -O2 -I include tests/native/gather_test.cpp -L vg-renderer_amd -lvgx -L/opt/rocm/lib -lrccl -Wl,-rpath,... -o gather_test
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "vgx.h"

#define CHECK(call)                                                                                   \
	do {                                                                                              \
		const int st_ = (call);                                                                       \
		if (st_ != VGX_OK) {                                                                          \
			fprintf(stderr, "%s:%d %s failed: %s (%d)\n", __FILE__, __LINE__, #call, vgx_status_string(st_), st_); \
			exit(1);                                                                                  \
		}                                                                                             \
	} while (0)
#define HIPOK(call) do { if ((call) != hipSuccess) { fprintf(stderr, "%s:%d %s failed\n", __FILE__, __LINE__, #call); exit(1); } } while (0)

static const uint8_t kCmdType[] = { VGX_CMD_MOVE_TO, VGX_CMD_CUBIC_TO, VGX_CMD_LINE_TO, VGX_CMD_CLOSE,   // path 0: closed blob
                                    VGX_CMD_MOVE_TO, VGX_CMD_LINE_TO, VGX_CMD_QUAD_TO,                     // path 1: open stroke
                                    VGX_CMD_MOVE_TO, VGX_CMD_LINE_TO, VGX_CMD_LINE_TO, VGX_CMD_CLOSE };    // path 2: triangle
static const uint32_t kArgOff[] = { 0, 2, 8, 10, 10, 12, 14, 18, 20, 22, 24, 24 };
static const float kArgs[] = { 0, 0, 22.5f, 0, 45, 22.5f, 45, 45, 0, 45,
                               5, 5, 60, 10, 80, 40, 30, 70,
                               0, 0, 30, 0, 15, 25 };
static const uint32_t kPathBegin[] = { 0, 4, 7, 11 };

static std::vector<vgx_draw> make_draws(uint64_t n)
{
	std::vector<vgx_draw> d(n);
	for (uint64_t i = 0; i < n; ++i) {
		vgx_draw r = {};
		r.path = (uint32_t)(i % 3);
		if (r.path != 1) { r.fill_flags = VGX_FILL_ENABLE | VGX_FILL_AA; r.fill_color = 0xFF204080u + (uint32_t)(i & 0xFF); }
		if (r.path != 2) { r.stroke_flags = VGX_STROKE_FLAGS(i % 2 ? VGX_CAP_ROUND : VGX_CAP_BUTT, i % 5 ? VGX_JOIN_MITER : VGX_JOIN_ROUND, 1, 0); r.stroke_color = 0xFF0000FFu; r.stroke_width = 3.0f + (float)(i % 4); }
		r.scale = 1.0f; r.tess_tol = 0.25f; r.fringe = 1.0f;
		r.mtx[0] = 1.0f; r.mtx[3] = 1.0f; r.mtx[4] = 50.0f * (float)(i % 97); r.mtx[5] = 50.0f * (float)(i / 97);
		d[i] = r;
	}
	return d;
}

struct Streams { std::vector<float> pos; std::vector<uint32_t> color; std::vector<uint16_t> idx; std::vector<vgx_mesh> meshes; };

struct Tess // one context's result, kept on the device
{
	vgx_ctx* ctx; vgx_pathset* ps; vgx_draw* draws; vgx_mesh_out out; vgx_sizes sz;
};

static Tess tessellate(const std::vector<vgx_draw>& all, uint64_t lo, uint64_t hi, hipStream_t s)
{
	Tess t = {};
	CHECK(vgx_create(0, &t.ctx));
	vgx_pathset_desc desc = { kCmdType, kArgOff, kArgs, kPathBegin, 3, 11 };
	CHECK(vgx_pathset_create(t.ctx, &desc, &t.ps));
	const uint64_t n = hi - lo;
	HIPOK(hipMalloc(&t.draws, (n + 1) * sizeof(vgx_draw)));
	HIPOK(hipMemcpy(t.draws, all.data() + lo, n * sizeof(vgx_draw), hipMemcpyHostToDevice));
	CHECK(vgx_tessellate_count(t.ctx, t.ps, t.draws, n, &t.sz, s));
	t.out.cap_vertices = t.sz.num_vertices; t.out.cap_indices = t.sz.num_indices; t.out.cap_meshes = t.sz.num_meshes;
	HIPOK(hipMalloc(&t.out.pos, (t.sz.num_vertices + 1) * 8));
	HIPOK(hipMalloc(&t.out.color, (t.sz.num_vertices + 1) * 4));
	HIPOK(hipMalloc(&t.out.idx, (t.sz.num_indices + 1) * 2));
	HIPOK(hipMalloc(&t.out.meshes, (t.sz.num_meshes + 1) * sizeof(vgx_mesh)));
	CHECK(vgx_tessellate_emit(t.ctx, t.ps, t.draws, n, &t.out, s));
	return t;
}

static Streams download(const vgx_mesh_out& o, uint64_t nv, uint64_t ni, uint64_t nm)
{
	Streams h;
	h.pos.resize(nv * 2); h.color.resize(nv); h.idx.resize(ni); h.meshes.resize(nm);
	HIPOK(hipMemcpy(h.pos.data(), o.pos, nv * 8, hipMemcpyDeviceToHost));
	HIPOK(hipMemcpy(h.color.data(), o.color, nv * 4, hipMemcpyDeviceToHost));
	HIPOK(hipMemcpy(h.idx.data(), o.idx, ni * 2, hipMemcpyDeviceToHost));
	HIPOK(hipMemcpy(h.meshes.data(), o.meshes, nm * sizeof(vgx_mesh), hipMemcpyDeviceToHost));
	return h;
}

static int g_fail = 0;

// a rank's draws [lo, hi) cut into T tiles; tile t tessellated into the rank's local buffers behind tile t - 1, then gathered
static void rank_tiles(int rank, int nranks, int root, int T, void* comm, const std::vector<vgx_draw>* all, const Streams* ref, const vgx_sizes* refSz,
                       std::vector<std::vector<vgx_rank_sizes>>* tileSizes /* [tile][rank], filled by the ranks, read after a barrier */, void (*barrier)(void*), void* barrierArg)
{
	HIPOK(hipSetDevice(0));
	hipStream_t s;
	HIPOK(hipStreamCreate(&s));
	const uint64_t n = all->size();
	const uint64_t base = n / nranks, rem = n % nranks;
	const uint64_t lo = rank * base + ((uint64_t)rank < rem ? rank : rem), hi = lo + base + ((uint64_t)rank < rem ? 1 : 0);
	// tessellate every tile with its own context (count first: sizes), into one set of local buffers
	std::vector<Tess> tiles;
	std::vector<uint64_t> tlo(T + 1);
	for (int t = 0; t <= T; ++t) { tlo[t] = lo + (hi - lo) * (uint64_t)t / (uint64_t)T; }
	for (int t = 0; t < T; ++t) {
		tiles.push_back(tessellate(*all, tlo[t], tlo[t + 1], s));
		(*tileSizes)[t][rank] = vgx_rank_sizes{ tiles[t].sz.num_vertices, tiles[t].sz.num_indices, tiles[t].sz.num_meshes, tlo[t + 1] - tlo[t] };
	}
	HIPOK(hipStreamSynchronize(s));
	barrier(barrierArg); // everybody's tile sizes are known (a real host would all-gather them: vgx_gather_sizes per tile)
	// rank bases = everything of the ranks in front; tile places = rank base + my tiles in front
	std::vector<vgx_rank_sizes> rankBase(nranks), sum(nranks);
	vgx_rank_sizes run = { 0, 0, 0, 0 };
	for (int r = 0; r < nranks; ++r) {
		rankBase[r] = run;
		for (int t = 0; t < T; ++t) { const vgx_rank_sizes& z = (*tileSizes)[t][r]; run.num_vertices += z.num_vertices; run.num_indices += z.num_indices; run.num_meshes += z.num_meshes; run.num_draws += z.num_draws; }
	}
	vgx_mesh_out g = {};
	if (rank == root) {
		g.cap_vertices = run.num_vertices; g.cap_indices = run.num_indices; g.cap_meshes = run.num_meshes;
		if (run.num_vertices != refSz->num_vertices || run.num_indices != refSz->num_indices || run.num_meshes != refSz->num_meshes) { fprintf(stderr, "tile sizes do not add up to the single-context run\n"); g_fail = 1; }
		HIPOK(hipMalloc(&g.pos, (g.cap_vertices + 1) * 8)); HIPOK(hipMalloc(&g.color, (g.cap_vertices + 1) * 4)); HIPOK(hipMalloc(&g.idx, (g.cap_indices + 1) * 2)); HIPOK(hipMalloc(&g.meshes, (g.cap_meshes + 1) * sizeof(vgx_mesh)));
	}
	std::vector<vgx_rank_sizes> place(nranks), front(nranks, vgx_rank_sizes{ 0, 0, 0, 0 });
	for (int t = 0; t < T; ++t) {
		for (int r = 0; r < nranks; ++r) {
			place[r] = vgx_rank_sizes{ rankBase[r].num_vertices + front[r].num_vertices, rankBase[r].num_indices + front[r].num_indices,
			                           rankBase[r].num_meshes + front[r].num_meshes, rankBase[r].num_draws + front[r].num_draws };
		}
		CHECK(vgx_gather_at(tiles[t].ctx, comm, root, &tiles[t].out, (*tileSizes)[t].data(), place.data(), rank == root ? &g : nullptr, s));
		for (int r = 0; r < nranks; ++r) { const vgx_rank_sizes& z = (*tileSizes)[t][r]; front[r].num_vertices += z.num_vertices; front[r].num_indices += z.num_indices; front[r].num_meshes += z.num_meshes; front[r].num_draws += z.num_draws; }
	}
	HIPOK(hipStreamSynchronize(s));
	if (rank == root) {
		const Streams got = download(g, refSz->num_vertices, refSz->num_indices, refSz->num_meshes);
		if (memcmp(got.pos.data(), ref->pos.data(), got.pos.size() * 4) != 0) { fprintf(stderr, "tiles: positions differ\n"); g_fail = 1; }
		if (memcmp(got.color.data(), ref->color.data(), got.color.size() * 4) != 0) { fprintf(stderr, "tiles: colours differ\n"); g_fail = 1; }
		if (memcmp(got.idx.data(), ref->idx.data(), got.idx.size() * 2) != 0) { fprintf(stderr, "tiles: indices differ\n"); g_fail = 1; }
		if (memcmp(got.meshes.data(), ref->meshes.data(), got.meshes.size() * sizeof(vgx_mesh)) != 0) { fprintf(stderr, "tiles: mesh tables differ\n"); g_fail = 1; }
		printf("ranks %d root %d, %d tiles per rank (vgx_gather_at): %llu vertices gathered, %s\n", nranks, root, T, (unsigned long long)refSz->num_vertices,
		       g_fail ? "MISMATCH" : "identical to the single-context run");
	}
}

// N ranks send the same local block: the destination is N times its size
static void rank_big(int rank, int nranks, void* comm, const Tess* shared, vgx_mesh_out* g)
{
	HIPOK(hipSetDevice(0));
	hipStream_t s;
	HIPOK(hipStreamCreate(&s));
	vgx_ctx* ctx;
	CHECK(vgx_create(0, &ctx));
	std::vector<vgx_rank_sizes> allSz(nranks, vgx_rank_sizes{ shared->sz.num_vertices, shared->sz.num_indices, shared->sz.num_meshes, 1000 });
	CHECK(vgx_gather(ctx, comm, 0, &shared->out, allSz.data(), rank == 0 ? g : nullptr, s));
	HIPOK(hipStreamSynchronize(s));
	CHECK(vgx_destroy(ctx));
}

static void rank_main(int rank, int nranks, int root, void* comm, const std::vector<vgx_draw>* all, const Streams* ref, const vgx_sizes* refSz)
{
	HIPOK(hipSetDevice(0));
	hipStream_t s;
	HIPOK(hipStreamCreate(&s));
	const uint64_t n = all->size();
	const uint64_t base = n / nranks, rem = n % nranks;
	const uint64_t lo = rank * base + ((uint64_t)rank < rem ? rank : rem), hi = lo + base + ((uint64_t)rank < rem ? 1 : 0);
	Tess t = tessellate(*all, lo, hi, s);
	vgx_rank_sizes mine = { t.sz.num_vertices, t.sz.num_indices, t.sz.num_meshes, hi - lo };
	std::vector<vgx_rank_sizes> allSz(nranks);
	CHECK(vgx_gather_sizes(t.ctx, comm, &mine, allSz.data(), s));
	vgx_mesh_out g = {};
	if (rank == root) {
		uint64_t tv = 0, ti = 0, tm = 0, td = 0;
		for (const vgx_rank_sizes& z : allSz) { tv += z.num_vertices; ti += z.num_indices; tm += z.num_meshes; td += z.num_draws; }
		if (tv != refSz->num_vertices || ti != refSz->num_indices || tm != refSz->num_meshes || td != n) { fprintf(stderr, "gathered sizes differ from the single-context run\n"); g_fail = 1; }
		g.cap_vertices = tv; g.cap_indices = ti; g.cap_meshes = tm;
		HIPOK(hipMalloc(&g.pos, (tv + 1) * 8)); HIPOK(hipMalloc(&g.color, (tv + 1) * 4)); HIPOK(hipMalloc(&g.idx, (ti + 1) * 2)); HIPOK(hipMalloc(&g.meshes, (tm + 1) * sizeof(vgx_mesh)));
		// a too small destination must be refused before anything is posted (the other ranks are not involved yet)
		vgx_mesh_out small = g; small.cap_vertices = tv ? tv - 1 : 0;
		if (nranks == 1 && tv && vgx_gather(t.ctx, comm, root, &t.out, allSz.data(), &small, s) != VGX_E_NOSPACE) { fprintf(stderr, "capacity check missing\n"); g_fail = 1; }
	}
	CHECK(vgx_gather(t.ctx, comm, root, &t.out, allSz.data(), rank == root ? &g : nullptr, s));
	HIPOK(hipStreamSynchronize(s));
	if (rank == root) {
		const Streams got = download(g, refSz->num_vertices, refSz->num_indices, refSz->num_meshes);
		if (memcmp(got.pos.data(), ref->pos.data(), got.pos.size() * 4) != 0) { fprintf(stderr, "positions differ\n"); g_fail = 1; }
		if (memcmp(got.color.data(), ref->color.data(), got.color.size() * 4) != 0) { fprintf(stderr, "colours differ\n"); g_fail = 1; }
		if (memcmp(got.idx.data(), ref->idx.data(), got.idx.size() * 2) != 0) { fprintf(stderr, "indices differ\n"); g_fail = 1; }
		if (memcmp(got.meshes.data(), ref->meshes.data(), got.meshes.size() * sizeof(vgx_mesh)) != 0) { fprintf(stderr, "mesh tables differ\n"); g_fail = 1; }
		printf("ranks %d root %d: %llu vertices, %llu indices, %llu meshes gathered, %s\n", nranks, root, (unsigned long long)refSz->num_vertices,
		       (unsigned long long)refSz->num_indices, (unsigned long long)refSz->num_meshes, g_fail ? "MISMATCH" : "identical to the single-context run");
	}
}

struct Barrier { std::mutex m; std::condition_variable cv; int n, arrived, gen; };
static void barrier_wait(void* p)
{
	Barrier* b = (Barrier*)p;
	std::unique_lock<std::mutex> l(b->m);
	const int gen = b->gen;
	if (++b->arrived == b->n) { b->arrived = 0; ++b->gen; b->cv.notify_all(); } else { b->cv.wait(l, [&] { return b->gen != gen; }); }
}

int main(int argc, char** argv)
{
	const bool tilesMode = argc > 5 && strcmp(argv[1], "tiles") == 0;
	const bool bigMode = argc > 4 && strcmp(argv[1], "big") == 0;
	if (bigMode) {
		const int nranks = atoi(argv[3]);
		const double gb = atof(argv[4]);
		HIPOK(hipSetDevice(0));
		setenv("VGX_RCCL_LIB", argv[2], 1);
		void* h = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
		if (!h) { fprintf(stderr, "dlopen %s: %s\n", argv[2], dlerror()); return 1; }
		void* (*mkShared)(int) = (void* (*)(int))dlsym(h, "fake_rccl_shared_create");
		void* (*mkComm)(void*, int) = (void* (*)(void*, int))dlsym(h, "fake_rccl_comm_create");
		// about 21 output bytes per vertex + 2 per index: pick the draw count from a probe batch
		const std::vector<vgx_draw> probe = make_draws(3000);
		Tess pt = tessellate(probe, 0, probe.size(), nullptr);
		HIPOK(hipDeviceSynchronize());
		const double bytesPerDraw = (12.0 * pt.sz.num_vertices + 2.0 * pt.sz.num_indices + 32.0 * pt.sz.num_meshes) / 3000.0;
		const uint64_t ndraws = (uint64_t)(gb * 1e9 / bytesPerDraw);
		const std::vector<vgx_draw> all = make_draws(ndraws);
		Tess one = tessellate(all, 0, all.size(), nullptr);
		HIPOK(hipDeviceSynchronize());
		const uint64_t nv = one.sz.num_vertices, ni = one.sz.num_indices, nm = one.sz.num_meshes;
		vgx_mesh_out g = {};
		g.cap_vertices = nv * nranks; g.cap_indices = ni * nranks; g.cap_meshes = nm * nranks;
		HIPOK(hipMalloc(&g.pos, (g.cap_vertices + 1) * 8)); HIPOK(hipMalloc(&g.color, (g.cap_vertices + 1) * 4)); HIPOK(hipMalloc(&g.idx, (g.cap_indices + 1) * 2)); HIPOK(hipMalloc(&g.meshes, (g.cap_meshes + 1) * sizeof(vgx_mesh)));
		void* shared = mkShared(nranks);
		std::vector<std::thread> th;
		for (int r = 0; r < nranks; ++r) { th.emplace_back(rank_big, r, nranks, mkComm(shared, r), &one, &g); }
		for (std::thread& t : th) { t.join(); }
		// every rank's block = the local block; its mesh records rebased by r blocks
		std::vector<uint8_t> a(1 << 20), b(1 << 20);
		for (int r = 0; r < nranks; ++r) {
			for (int where = 0; where < 2; ++where) { // first and last megabyte of the position / index blocks
				const size_t pb = nv * 8, ib = ni * 2;
				const size_t po = where ? (pb > a.size() ? pb - a.size() : 0) : 0, io = where ? (ib > a.size() ? ib - a.size() : 0) : 0;
				const size_t pn = pb - po < a.size() ? pb - po : a.size(), in = ib - io < a.size() ? ib - io : a.size();
				HIPOK(hipMemcpy(a.data(), (const uint8_t*)one.out.pos + po, pn, hipMemcpyDeviceToHost));
				HIPOK(hipMemcpy(b.data(), (const uint8_t*)g.pos + (size_t)r * pb + po, pn, hipMemcpyDeviceToHost));
				if (memcmp(a.data(), b.data(), pn) != 0) { fprintf(stderr, "big: positions of rank %d differ\n", r); g_fail = 1; }
				HIPOK(hipMemcpy(a.data(), (const uint8_t*)one.out.idx + io, in, hipMemcpyDeviceToHost));
				HIPOK(hipMemcpy(b.data(), (const uint8_t*)g.idx + (size_t)r * ib + io, in, hipMemcpyDeviceToHost));
				if (memcmp(a.data(), b.data(), in) != 0) { fprintf(stderr, "big: indices of rank %d differ\n", r); g_fail = 1; }
			}
			vgx_mesh first, last, lfirst, llast;
			HIPOK(hipMemcpy(&first, g.meshes + (size_t)r * nm, sizeof(vgx_mesh), hipMemcpyDeviceToHost));
			HIPOK(hipMemcpy(&last, g.meshes + (size_t)r * nm + nm - 1, sizeof(vgx_mesh), hipMemcpyDeviceToHost));
			HIPOK(hipMemcpy(&lfirst, one.out.meshes, sizeof(vgx_mesh), hipMemcpyDeviceToHost));
			HIPOK(hipMemcpy(&llast, one.out.meshes + nm - 1, sizeof(vgx_mesh), hipMemcpyDeviceToHost));
			if (first.first_vertex != lfirst.first_vertex + (uint64_t)r * nv || last.first_index != llast.first_index + (uint64_t)r * ni || last.draw != llast.draw + (uint32_t)(r * 1000)) {
				fprintf(stderr, "big: mesh records of rank %d not rebased\n", r); g_fail = 1;
			}
		}
		printf("big: %d ranks x %.2f GB = %.2f GB gathered into one destination (%llu vertices, %llu indices), %s\n", nranks,
		       (12.0 * nv + 2.0 * ni + 32.0 * nm) / 1e9, nranks * (12.0 * nv + 2.0 * ni + 32.0 * nm) / 1e9, (unsigned long long)g.cap_vertices, (unsigned long long)g.cap_indices, g_fail ? "MISMATCH" : "blocks identical, mesh records rebased");
		return g_fail;
	}
	const bool fake = argc > 1 && (strcmp(argv[1], "fake") == 0 || tilesMode);
	const int nranks = fake ? atoi(argv[3]) : 1;
	const uint64_t ndraws = (fake && argc > 4) ? strtoull(argv[4], nullptr, 10) : 1000; // fewer draws than ranks: empty shards
	HIPOK(hipSetDevice(0));
	const std::vector<vgx_draw> all = make_draws(ndraws);
	Tess whole = tessellate(all, 0, all.size(), nullptr);
	HIPOK(hipDeviceSynchronize());
	const Streams ref = download(whole.out, whole.sz.num_vertices, whole.sz.num_indices, whole.sz.num_meshes);
	if (!fake) {
		ncclUniqueId id;
		ncclComm_t comm;
		if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) { fprintf(stderr, "ncclCommInitRank failed\n"); return 1; }
		rank_main(0, 1, 0, comm, &all, &ref, &whole.sz);
		ncclCommDestroy(comm);
		return g_fail;
	}
	setenv("VGX_RCCL_LIB", argv[2], 1);
	void* h = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
	if (!h) { fprintf(stderr, "dlopen %s: %s\n", argv[2], dlerror()); return 1; }
	void* (*mkShared)(int) = (void* (*)(int))dlsym(h, "fake_rccl_shared_create");
	void* (*mkComm)(void*, int) = (void* (*)(void*, int))dlsym(h, "fake_rccl_comm_create");
	if (tilesMode) {
		const int T = atoi(argv[5]);
		for (int root : { 0, nranks - 1 }) {
			void* shared = mkShared(nranks);
			std::vector<std::vector<vgx_rank_sizes>> tileSizes(T, std::vector<vgx_rank_sizes>(nranks));
			Barrier bar; bar.n = nranks; bar.arrived = 0; bar.gen = 0;
			std::vector<std::thread> th;
			for (int r = 0; r < nranks; ++r) { th.emplace_back(rank_tiles, r, nranks, root, T, mkComm(shared, r), &all, &ref, &whole.sz, &tileSizes, barrier_wait, &bar); }
			for (std::thread& t : th) { t.join(); }
		}
		return g_fail;
	}
	for (int root : { 0, nranks - 1 }) {
		void* shared = mkShared(nranks);
		std::vector<std::thread> th;
		for (int r = 0; r < nranks; ++r) { th.emplace_back(rank_main, r, nranks, root, mkComm(shared, r), &all, &ref, &whole.sz); }
		for (std::thread& t : th) { t.join(); }
	}
	return g_fail;
}
