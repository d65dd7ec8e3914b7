"""Tuning probe: does running TWO half batches on two streams (two contexts) overlap the VALU-bound flatten of one half
with the memory-bound emit of the other? Prints ms for: whole batch on one stream; two halves back to back on one stream;
two halves on two streams; four quarters on two / four streams."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

rt = importlib.import_module("vg-renderer_amd.runtime")
wl = importlib.import_module("vg-renderer_amd.workloads")
K = 10000
ps, ops = wl.tiger_paths()


class Part:
    def __init__(self, first, n):
        self.ctx = rt.Context(0)
        self.pset = rt.PathSet(self.ctx, ps)
        d = wl.tiger_draws(ops, n, first_instance=first)
        self.n = d.shape[0]
        self.dd = rt.upload_draws(d, 0)
        sizes = rt.tessellate_count(self.ctx, self.pset, self.dd, self.n)
        self.bufs = rt.MeshBuffers(torch.device("cuda", 0), sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
        self.stream = torch.cuda.Stream()

    def run(self, stream=None):
        with torch.cuda.stream(stream or self.stream):
            rt.tessellate_async(self.ctx, self.pset, self.dd, self.n, self.bufs)


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


whole = Part(0, K)
print("whole batch, one stream: %.3f ms" % timeit(lambda: whole.run()))
del whole
torch.cuda.empty_cache()
for parts in (2, 4, 8):
    P = [Part(i * (K // parts), K // parts) for i in range(parts)]
    s0 = torch.cuda.Stream()
    print("%d parts, one stream:        %.3f ms" % (parts, timeit(lambda: [p.run(s0) for p in P])))
    print("%d parts, one stream each:   %.3f ms" % (parts, timeit(lambda: [p.run() for p in P])))
    if parts > 2:
        ss = [torch.cuda.Stream(), torch.cuda.Stream()]
        print("%d parts, two streams:       %.3f ms" % (parts, timeit(lambda: [p.run(ss[i % 2]) for i, p in enumerate(P)])))
    del P
    torch.cuda.empty_cache()

# staggered: the second stream starts one flatten-time later, so that flatten(B) runs beside emit(A)
for parts in (2, 4, 8):
    P = [Part(i * (K // parts), K // parts) for i in range(parts)]
    ss = [torch.cuda.Stream(), torch.cuda.Stream()]
    for delay_ms in (0.0, 2.6 / parts * 0.5, 2.6 / parts, 2.6 / parts * 1.5):
        cycles = int(delay_ms * 1e-3 * 100e6 * 21)  # torch.cuda._sleep counts ~clock cycles; calibrated below

        def go():
            if cycles:
                with torch.cuda.stream(ss[1]):
                    torch.cuda._sleep(cycles)
            for i, p in enumerate(P):
                p.run(ss[i % 2])
        print("%d parts, two streams, second delayed by ~%.2f ms: %.3f ms" % (parts, delay_ms, timeit(go)))
    del P
    torch.cuda.empty_cache()
t0 = time.perf_counter(); torch.cuda._sleep(int(1e-3 * 100e6 * 21)); torch.cuda.synchronize(); print("sleep calibration: nominal 1 ms took %.3f ms" % ((time.perf_counter() - t0) * 1e3))
