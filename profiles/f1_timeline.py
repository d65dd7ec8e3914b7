"""Timeline of vgx_flatten's tickets from a -DVGX_F1_PROFILE build: per ticket the 100 MHz wall clock at the ticket, at A(t) (walk +
bookkeeping done) and after the look-back, + the hardware id. Prints how late the predecessors were and where stragglers sit."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
rt = importlib.import_module("vg-renderer_amd.runtime")
wl = importlib.import_module("vg-renderer_amd.workloads")
ps, d = wl.random_cubics(1000000, seed=1234, box=1000.0)
ctx = rt.Context(0)
pset = rt.PathSet(ctx, ps)
dd = rt.upload_draws(d)
r = rt.flatten(ctx, pset, dd, d.shape[0], entry="two_phase", to_host=False)
fb = rt.FlatBuffers(dd.device, r.sizes["num_poly_vertices"], r.sizes["num_subpaths"], d.shape[0])
for _ in range(3):
    rt.flatten_async(ctx, pset, dd, d.shape[0], fb, apply_transform=True)
torch.cuda.synchronize()
n = 31250
dbg = torch.zeros((n, 4), dtype=torch.int64, device=dd.device)
L = rt.lib()
L.vgx_f1_debug_buffer.argtypes = [C.c_void_p, C.c_uint64]
assert L.vgx_f1_debug_buffer(dbg.data_ptr(), n) == 0
rt.flatten_async(ctx, pset, dd, d.shape[0], fb, apply_transform=True)
torch.cuda.synchronize()
L.vgx_f1_debug_buffer(None, 0)
a = dbg.cpu().numpy().astype(np.int64)
t0 = a[:, 0].min()
tk, ta, tb, hw = a[:, 0] - t0, a[:, 1] - t0, a[:, 2] - t0, a[:, 3]
print("kernel span %.1f us; per ticket: front (ticket -> A) mean %.1f us (p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f); wait (A -> base) mean %.1f us p50 %.1f p90 %.1f" % (
    (tb.max()) / 100.0, (ta - tk).mean() / 100.0, *[np.percentile(ta - tk, q) / 100.0 for q in (10, 50, 90, 99, 100)], (tb - ta).mean() / 100.0, np.percentile(tb - ta, 50) / 100.0, np.percentile(tb - ta, 90) / 100.0))
pm = np.maximum.accumulate(ta)  # latest A among tickets <= k
late = np.concatenate([[0], pm[:-1]]) - ta  # how much later the slowest predecessor published than I did
print("slowest predecessor later than me by: mean %.1f us, p50 %.1f, p90 %.1f; tickets whose own A set a new maximum: %d of %d" % (late.clip(0).mean() / 100.0, np.percentile(late, 50) / 100.0, np.percentile(late, 90) / 100.0, int((ta >= pm).sum()), n))
order = np.argsort(tk, kind="stable")
print("ticket times monotonic in ticket order: %s; ticket spacing mean %.3f us" % (bool((np.diff(tk) >= -2).all()), float(np.diff(np.sort(tk)).mean()) / 100.0))
front = ta - tk
deepf = (hw >> 63) & 1
overf = (hw >> 62) & 1
hw = hw & ((1 << 62) - 1)
print("tickets with a full-depth redo: %d (front mean %.1f us), with a list overflow: %d; front of the others: mean %.1f us p99 %.1f max %.1f" % (
    int(deepf.sum()), float(front[deepf == 1].mean()) / 100.0 if deepf.any() else 0.0, int(overf.sum()), float(front[deepf == 0].mean()) / 100.0,
    np.percentile(front[deepf == 0], 99) / 100.0, front[deepf == 0].max() / 100.0))
xcc = (hw >> 0) & 0xFFFFFFFF
hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF; se = (hwid >> 13) & 0x7; sh = (hwid >> 12) & 1; simd = (hwid >> 4) & 0x3; wave = hwid & 0xF
blk = hw >> 32
print("front time by SIMD id:", [round(float(front[simd == k].mean()) / 100.0, 1) for k in range(4)])
print("front time by SE id:", [round(float(front[se == k].mean()) / 100.0, 1) if (se == k).any() else None for k in range(8)])
slow = np.argsort(front)[-20:]
print("20 slowest fronts (us, ticket, block, se, cu, simd):", [(round(front[i] / 100.0, 1), int(i), int(blk[i]), int(se[i]), int(cu[i]), int(simd[i])) for i in slow])
# generations: tickets per wave in sequence
for b in (0, 1, 777):
    idx = np.flatnonzero(blk == b)
    print("block %d: tickets %s ... front us %s wait us %s" % (b, idx[:6].tolist(), (front[idx[:6]] / 100.0).round(1).tolist(), ((tb - ta)[idx[:6]] / 100.0).round(1).tolist()))
