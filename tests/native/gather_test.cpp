// gather_test.cpp -- vgx_gather_sizes / vgx_gather driven from C++ the way a multi-GPU host would (one context + one
// communicator per rank), checked against ONE context tessellating the whole batch: the gathered streams must be
// byte-identical and the gathered mesh table must equal the single-context table (first_vertex / first_index / draw
// rebased by the ranks in front).
//   gather_test real            1 rank, the real librccl (ncclCommInitRank with nranks = 1): binding, collective, copies
//   gather_test fake LIB N      N ranks as threads of this process on one GPU over tests/native/fake_rccl.cpp (VGX_RCCL_LIB=LIB)
// Build: hipcc -O2 -I include tests/native/gather_test.cpp -L vg-renderer_amd -lvgx -L/opt/rocm/lib -lrccl -Wl,-rpath,... -o gather_test
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

int main(int argc, char** argv)
{
	const bool fake = argc > 1 && strcmp(argv[1], "fake") == 0;
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
	for (int root : { 0, nranks - 1 }) {
		void* shared = mkShared(nranks);
		std::vector<std::thread> th;
		for (int r = 0; r < nranks; ++r) { th.emplace_back(rank_main, r, nranks, root, mkComm(shared, r), &all, &ref, &whole.sz); }
		for (std::thread& t : th) { t.join(); }
	}
	return g_fail;
}
