// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for the seven RCCL entry points vgx_gather binds, for "ranks" that are
// THREADS of one process sharing one GPU (RCCL itself refuses two ranks on one device, and the GPU test boxes have one).
// Point-to-point = a mailbox per (source, destination) pair + a device-to-device copy on the receiver's stream; the
// collective = a barrier + copies. It lets tests/native/gather_test.cpp run vgx_gather_sizes / vgx_gather with 2-4 ranks and
// check offsets, capacities and the mesh-table rebase against a single-context run. libvgx binds it through
// VGX_RCCL_LIB (testing knob); the real library is exercised by the 1-rank leg of the same test and by the driver's
// multi-GPU bench.
//   g++ -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/native/fake_rccl.cpp -L/opt/rocm/lib -lamdhip64 -o libfake_rccl.so
#include <hip/hip_runtime_api.h>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <vector>
#include <stdint.h>
#include <stddef.h>

namespace {
struct Msg { const void* p; size_t bytes; };
struct Shared
{
	int nranks;
	std::mutex m;
	std::condition_variable cv;
	std::vector<std::deque<Msg>> box; // [src * nranks + dst]
	std::vector<const void*> slot;     // all-gather: every rank's send buffer
	int arrived, generation;
};
struct Comm { Shared* sh; int rank; };
size_t elemBytes(int dt) { return (dt == 4 || dt == 5 || dt == 8) ? 8 : ((dt == 2 || dt == 3 || dt == 7) ? 4 : (dt == 6 || dt == 9 ? 2 : 1)); }
void barrier(Shared* sh)
{
	std::unique_lock<std::mutex> l(sh->m);
	const int gen = sh->generation;
	if (++sh->arrived == sh->nranks) { sh->arrived = 0; ++sh->generation; sh->cv.notify_all(); }
	else { sh->cv.wait(l, [&] { return sh->generation != gen; }); }
}
}

extern "C" {
// test-side constructors (not part of RCCL)
void* fake_rccl_shared_create(int nranks)
{
	Shared* s = new Shared;
	s->nranks = nranks; s->box.resize((size_t)nranks * nranks); s->slot.resize(nranks); s->arrived = 0; s->generation = 0;
	return s;
}
void* fake_rccl_comm_create(void* shared, int rank) { Comm* c = new Comm; c->sh = (Shared*)shared; c->rank = rank; return c; }

int ncclGroupStart() { return 0; }
int ncclGroupEnd() { return 0; }
int ncclCommCount(void* comm, int* n) { *n = ((Comm*)comm)->sh->nranks; return 0; }
int ncclCommUserRank(void* comm, int* r) { *r = ((Comm*)comm)->rank; return 0; }
int ncclSend(const void* buf, size_t count, int dt, int peer, void* comm, hipStream_t s)
{
	Comm* c = (Comm*)comm;
	if (hipStreamSynchronize(s) != hipSuccess) { return 1; } // the data is final before it is announced
	std::lock_guard<std::mutex> l(c->sh->m);
	c->sh->box[(size_t)c->rank * c->sh->nranks + peer].push_back(Msg{ buf, count * elemBytes(dt) });
	c->sh->cv.notify_all();
	return 0;
}
int ncclRecv(void* buf, size_t count, int dt, int peer, void* comm, hipStream_t s)
{
	Comm* c = (Comm*)comm;
	Msg m;
	{
		std::unique_lock<std::mutex> l(c->sh->m);
		std::deque<Msg>& q = c->sh->box[(size_t)peer * c->sh->nranks + c->rank];
		c->sh->cv.wait(l, [&] { return !q.empty(); });
		m = q.front(); q.pop_front();
	}
	if (m.bytes != count * elemBytes(dt)) { return 5; } // ncclInvalidArgument: send / receive sizes must match
	return hipMemcpyAsync(buf, m.p, m.bytes, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : 1;
}
int ncclAllGather(const void* sendbuf, void* recvbuf, size_t count, int dt, void* comm, hipStream_t s)
{
	Comm* c = (Comm*)comm;
	if (hipStreamSynchronize(s) != hipSuccess) { return 1; }
	{ std::lock_guard<std::mutex> l(c->sh->m); c->sh->slot[c->rank] = sendbuf; }
	barrier(c->sh);
	const size_t bytes = count * elemBytes(dt);
	for (int r = 0; r < c->sh->nranks; ++r) {
		void* dst = (char*)recvbuf + (size_t)r * bytes;
		if (dst != c->sh->slot[r] && hipMemcpyAsync(dst, c->sh->slot[r], bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { return 1; }
	}
	if (hipStreamSynchronize(s) != hipSuccess) { return 1; }
	barrier(c->sh); // nobody reuses its send buffer before everybody has copied it
	return 0;
}
}
