// h2d_probe.hip -- how fast do a path set's raw arrays reach HBM from ordinary (pageable) host memory? (round 6, vgx_pathset_create)
//   hipcc --offload-arch=gfx950 -O2 -o h2d_probe.bin h2d_probe.hip -lpthread && ./h2d_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static void par_memcpy(char* d, const char* s, size_t n, int threads)
{
	if (threads <= 1) { memcpy(d, s, n); return; }
	std::vector<std::thread> th;
	const size_t per = (n + threads - 1) / threads;
	for (int t = 0; t < threads; ++t) { const size_t o = per * t; if (o >= n) break; const size_t m = n - o < per ? n - o : per; th.emplace_back([=] { memcpy(d + o, s + o, m); }); }
	for (auto& t : th) t.join();
}
int main()
{
	hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	for (size_t mb : {46, 130}) {
		const size_t n = mb << 20;
		char* src = (char*)malloc(n); memset(src, 1, n);
		void* dst; CK(hipMalloc(&dst, n));
		for (int rep = 0; rep < 2; ++rep) {
			double t0 = now(); CK(hipMemcpy(dst, src, n, hipMemcpyHostToDevice)); double t1 = now();
			printf("%3zu MB pageable hipMemcpy            %7.2f ms %6.1f GB/s\n", mb, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
		}
		for (size_t chunkMB : {2, 8, 32}) for (int threads : {1, 2, 4, 8}) {
			const size_t chunk = chunkMB << 20;
			char* st[2]; hipEvent_t ev[2];
			for (int i = 0; i < 2; ++i) { CK(hipHostMalloc((void**)&st[i], chunk, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); memset(st[i], 0, chunk); }
			double best = 1e9;
			for (int rep = 0; rep < 3; ++rep) {
				double t0 = now();
				size_t k = 0;
				for (size_t o = 0; o < n; o += chunk, ++k) {
					const size_t m = n - o < chunk ? n - o : chunk; const int i = k & 1;
					if (k >= 2) CK(hipEventSynchronize(ev[i]));
					par_memcpy(st[i], src + o, m, threads);
					CK(hipMemcpyAsync((char*)dst + o, st[i], m, hipMemcpyHostToDevice, s)); CK(hipEventRecord(ev[i], s));
				}
				CK(hipStreamSynchronize(s));
				double t1 = now(); if (t1 - t0 < best) best = t1 - t0;
			}
			printf("%3zu MB staged chunk %2zu MB x %d threads   %7.2f ms %6.1f GB/s\n", mb, chunkMB, threads, best * 1e3, n / best / 1e9);
			for (int i = 0; i < 2; ++i) { CK(hipHostFree(st[i])); CK(hipEventDestroy(ev[i])); }
		}
		{
			double t0 = now(); CK(hipHostRegister(src, n, hipHostRegisterDefault)); double t1 = now();
			CK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t2 = now();
			CK(hipHostUnregister(src)); double t3 = now();
			printf("%3zu MB hipHostRegister %6.2f + copy %6.2f (%5.1f GB/s) + unregister %6.2f = %7.2f ms\n", mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
		}
		{
			char* pin; CK(hipHostMalloc((void**)&pin, n, hipHostMallocDefault)); memset(pin, 2, n);
			double t0 = now(); CK(hipMemcpyAsync(dst, pin, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t1 = now();
			printf("%3zu MB pinned source                 %7.2f ms %6.1f GB/s\n", mb, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
			double m0 = now(); memcpy(pin, src, n); double m1 = now();
			printf("%3zu MB host memcpy (1 thread)        %7.2f ms %6.1f GB/s\n", mb, (m1 - m0) * 1e3, n / (m1 - m0) / 1e9);
			CK(hipHostFree(pin));
		}
		{ // hipMalloc / hipFree / memset of a blob of the size a path set of this many raw bytes needs (~8x)
			double t0 = now(); void* b; CK(hipMalloc(&b, n * 8)); double t1 = now(); CK(hipMemsetAsync(b, 0, n * 8, s)); CK(hipStreamSynchronize(s)); double t2 = now(); CK(hipFree(b)); double t3 = now();
			printf("%3zu MB x8 blob: hipMalloc %6.2f ms, memset %6.2f ms, hipFree %6.2f ms\n", mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
		}
		CK(hipFree(dst)); free(src);
	}
	return 0;
}
