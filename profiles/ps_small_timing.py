"""Host time of vgx_pathset_create / _destroy for a frame-sized set (the Tiger's 240 paths), and -- under rocprofv3 --kernel-trace --stats --
the duration of the one kernel that builds it.   python profiles/ps_small_timing.py [repeats]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("vg-renderer_amd.runtime")
wl = importlib.import_module("vg-renderer_amd.workloads")
import torch

R = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = rt.Context(0)
ps, _ = wl.tiger_paths()
for _ in range(5):
    rt.PathSet(ctx, ps).close()
torch.cuda.synchronize()
tc = td = 0.0
for _ in range(R):
    t0 = time.perf_counter()
    p = rt.PathSet(ctx, ps)
    t1 = time.perf_counter()
    p.close()
    t2 = time.perf_counter()
    tc += t1 - t0
    td += t2 - t1
print("tiger path set (%d commands): create %.1f us, destroy %.1f us (mean of %d)" % (ps.ncmd, tc / R * 1e6, td / R * 1e6, R))
