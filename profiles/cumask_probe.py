"""Tuning probe (round 3): do two streams with DISJOINT CU masks overlap the VALU / latency-bound front half of one tile
(scans, k_flatten_inst, gather) with the memory-bound emit half of another? Tiger x10k cut into `parts` tiles, tile i on
stream i % 2, the two streams restricted to complementary sets of CUs (hipExtStreamCreateWithCUMask)."""
import ctypes as C
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
hip = C.CDLL("libamdhip64.so")


def masked_stream(words):
    arr = (C.c_uint32 * len(words))(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


class Part:
    def __init__(self, first, n):
        self.ctx = rt.Context(0)
        self.pset = rt.PathSet(self.ctx, ps)
        d = wl.tiger_draws(ops, n, first_instance=first)
        self.n = d.shape[0]
        self.dd = rt.upload_draws(d, 0)
        sizes = rt.tessellate_count(self.ctx, self.pset, self.dd, self.n)
        self.bufs = rt.MeshBuffers(torch.device("cuda", 0), sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])

    def run(self, stream):
        with torch.cuda.stream(stream):
            rt.tessellate_async(self.ctx, self.pset, self.dd, self.n, self.bufs)


def timeit(fn, reps=6):
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


full = [0xFFFFFFFF] * 8
s_full = masked_stream(full)
for parts in (1, 4, 8, 16):
    P = [Part(i * (K // parts), K // parts) for i in range(parts)]
    s0 = torch.cuda.Stream()
    print("%2d parts, one stream:                         %.3f ms" % (parts, timeit(lambda: [p.run(s0) for p in P])), flush=True)
    if parts == 1:
        print("%2d parts, one stream with a full CU mask:     %.3f ms" % (parts, timeit(lambda: [p.run(s_full) for p in P])), flush=True)
        del P
        torch.cuda.empty_cache()
        continue
    for name, ma, mb in (("even / odd CUs", [0x55555555] * 8, [0xAAAAAAAA] * 8),
                         ("low / high half", [0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
                         ("both full masks", full, full)):
        sa, sb = masked_stream(ma), masked_stream(mb)
        print("%2d parts, two masked streams (%s): %.3f ms" % (parts, name, timeit(lambda: [p.run(sa if i % 2 == 0 else sb) for i, p in enumerate(P)])), flush=True)
    del P
    torch.cuda.empty_cache()
