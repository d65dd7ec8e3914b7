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
print('occupancy (blocks per CU) by cap:', {c: L.vgx_f1_debug_occupancy(c) for c in (1024, 1664, 2048, 3072)})
L.vgx_f1_debug_buffer.argtypes = [C.c_void_p, C.c_uint64]
assert L.vgx_f1_debug_buffer(dbg.data_ptr(), n) == 0
rt.flatten_async(ctx, pset, dd, d.shape[0], fb, apply_transform=True)
torch.cuda.synchronize()
L.vgx_f1_debug_buffer(None, 0)
a = dbg.cpu().numpy().astype(np.int64)
t0 = a[:, 0].min()
tk, ta, tb, hw = a[:, 0] - t0, a[:, 1] - t0, a[:, 2] - t0, a[:, 3]
tp = ((hw & ((1 << 40) - 1)) - (t0 & ((1 << 40) - 1))) % (1 << 40)   # wall clock at the end of the segment's front (command records decoded)
leaves = (hw >> 50) & 0xFFF
hwu = hw.astype(np.uint64)
hw = ((hwu & np.uint64(3 << 62)) | (((hwu >> np.uint64(40)) & np.uint64(0x3FF)) << np.uint64(32))).astype(np.uint64)
print("prologue (start -> decoded): mean %.1f us p50 %.1f p90 %.1f p99 %.1f max %.1f;  walk + bookkeeping (decoded -> A): mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
    (tp - tk).mean() / 100.0, *[np.percentile(tp - tk, q) / 100.0 for q in (50, 90, 99, 100)], (ta - tp).mean() / 100.0, *[np.percentile(ta - tp, q) / 100.0 for q in (50, 90, 99, 100)]))
print("leaves per chunk: mean %.0f sd %.0f p99 %d max %d" % (leaves.mean(), leaves.std(), np.percentile(leaves, 99), leaves.max()))
print("kernel span %.1f us; per ticket: front (ticket -> A) mean %.1f us (p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f); wait (A -> base) mean %.1f us p50 %.1f p90 %.1f" % (
    (tb.max()) / 100.0, (ta - tk).mean() / 100.0, *[np.percentile(ta - tk, q) / 100.0 for q in (10, 50, 90, 99, 100)], (tb - ta).mean() / 100.0, np.percentile(tb - ta, 50) / 100.0, np.percentile(tb - ta, 90) / 100.0))
pm = np.maximum.accumulate(ta)  # latest A among tickets <= k
late = np.concatenate([[0], pm[:-1]]) - ta  # how much later the slowest predecessor published than I did
print("slowest predecessor later than me by: mean %.1f us, p50 %.1f, p90 %.1f; tickets whose own A set a new maximum: %d of %d" % (late.clip(0).mean() / 100.0, np.percentile(late, 50) / 100.0, np.percentile(late, 90) / 100.0, int((ta >= pm).sum()), n))
order = np.argsort(tk, kind="stable")
print("ticket times monotonic in ticket order: %s; ticket spacing mean %.3f us" % (bool((np.diff(tk) >= -2).all()), float(np.diff(np.sort(tk)).mean()) / 100.0))
front = ta - tk
deepf = ((hw >> np.uint64(63)) & np.uint64(1)).astype(np.int64)
overf = ((hw >> np.uint64(62)) & np.uint64(1)).astype(np.int64)
hw = (hw & np.uint64((1 << 62) - 1)).astype(np.int64)
print("tickets with a full-depth redo: %d (front mean %.1f us), with a list overflow: %d; front of the others: mean %.1f us p99 %.1f max %.1f" % (
    int(deepf.sum()), float(front[deepf == 1].mean()) / 100.0 if deepf.any() else 0.0, int(overf.sum()), float(front[deepf == 0].mean()) / 100.0,
    np.percentile(front[deepf == 0], 99) / 100.0, front[deepf == 0].max() / 100.0))
hwid = np.zeros_like(hw)
cu = (hwid >> 8) & 0xF; se = (hwid >> 13) & 0x7; sh = (hwid >> 12) & 1; simd = (hwid >> 4) & 0x3; wave = hwid & 0xF
blk = hw >> 32
print("front time by SIMD id:", [round(float(front[simd == k].mean()) / 100.0, 1) for k in range(4)])
print("front time by SE id:", [round(float(front[se == k].mean()) / 100.0, 1) if (se == k).any() else None for k in range(8)])
slow = np.argsort(front)[-20:]
print("20 slowest fronts (us, ticket, block, se, cu, simd):", [(round(front[i] / 100.0, 1), int(i), int(blk[i]), int(se[i]), int(cu[i]), int(simd[i])) for i in slow])
# generations: tickets per wave in sequence
for b in (0, 1, 777, 1400):
    idx = np.flatnonzero(blk == b)
    print("block %d: %d tickets; first 8: %s\n   start us %s\n   A us     %s\n   base us  %s" % (b, idx.shape[0], idx[:8].tolist(), (tk[idx[:8]] / 100.0).round(1).tolist(), (ta[idx[:8]] / 100.0).round(1).tolist(), (tb[idx[:8]] / 100.0).round(1).tolist()))
    print("   last 4: %s start us %s base us %s" % (idx[-4:].tolist(), (tk[idx[-4:]] / 100.0).round(1).tolist(), (tb[idx[-4:]] / 100.0).round(1).tolist()))
nb = np.bincount(blk.astype(np.int64))
print("blocks that worked: %d; tickets per block min %d mean %.1f max %d" % (int((nb > 0).sum()), int(nb[nb > 0].min()), float(nb[nb > 0].mean()), int(nb.max())))
# concurrency over time: tickets in flight (between start and base) sampled every 50 us
for x in range(0, int(tb.max()), 10000):
    print("   t=%4d us: %d tickets between start and base" % (x // 100, int(((tk <= x) & (tb > x)).sum())))
