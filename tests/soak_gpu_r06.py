"""GPU soak (not collected by pytest) for round 6's kernels: N random seeds each through
  (a) k_emit_tiles: fuzz drawings of fills + closed Miter strokes, ordinary pipeline, the tile kernel armed for every call (VGX_BIG_EMIT_MIN=0);
  (b) k_stroke_long: batches of long polylines in random general styles (Round / Bevel / Miter joins, all caps, AA and not, closed and open);
  (c) vgx_pathset_create on the device: every table of random path sets (every command / lineTo-only) against the host loops (libvgx_hosttest.so);
  (d) every fourth seed: vgx_tessellate's one-walk flatten route on a fuzz path set of 2 100 - 3 200 paths (every command, shapes, degenerate draws);
all against the reference (oracle/_ref) bit for bit.   python tests/soak_gpu_r06.py 300"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyoracle
from util import assert_mesh_equal, run_async
rt = importlib.import_module("vg-renderer_amd.runtime"); wl = importlib.import_module("vg-renderer_amd.workloads"); vgr = importlib.import_module("vg-renderer_amd")
capi = rt.capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
BASE = int(sys.argv[2]) if len(sys.argv) > 2 else 6000


def ctx_with(**env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return rt.Context(0)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


bad = 0
ctx_tile = [ctx_with(VGX_TMPL=0, VGX_INST=0, VGX_BIG_EMIT_MIN=0), ctx_with(VGX_TMPL=0, VGX_INST=1, VGX_BIG_EMIT_MIN=0)]
ctx_long = ctx_with(VGX_BIG_EMIT_MIN=0)
ctx_f1 = ctx_with(VGX_TESS_FLAT1=2, VGX_BIG_EMIT_MIN=0)
hostlib = C.CDLL(os.path.join(ROOT, "vg-renderer_amd", "libvgx_hosttest.so"))
hostlib.vgxt_pathset_table.restype = C.c_int64
hostlib.vgxt_pathset_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
for seed in range(BASE, BASE + N):
    rs = np.random.RandomState(seed)
    try:
        # (a) tile kernel
        ps = wl.closed_fuzz_paths(seed, npaths=int(rs.randint(8, 72)))
        d = wl.template_draws(ps, seed, int(rs.randint(2, 40)))
        if seed % 3 == 0:
            d = d[rs.permutation(d.shape[0])]
        if seed % 5 == 0:
            d = d[rs.uniform(size=d.shape[0]) < 0.7]
        got = run_async(rt, ctx_tile[seed & 1], ps, d, profile=True)
        assert got.status == 0 and "tile_emit" in got.stages, got.stages
        assert_mesh_equal(got, pyoracle.tessellate(ps, d), "tile %d" % seed)
        # (b) long polylines
        cap = int(rs.choice([capi.CAP_BUTT, capi.CAP_ROUND, capi.CAP_SQUARE])); join = int(rs.choice([capi.JOIN_MITER, capi.JOIN_ROUND, capi.JOIN_BEVEL]))
        ps, d = wl.random_walk_polylines(int(rs.randint(3, 40)), int(rs.randint(130, 700)), seed=seed, width=float(rs.choice([0.6, 2.0, 6.0, 25.0, 80.0])), cap=cap, join=join,
                                         step=float(rs.choice([3.0, 8.0, 20.0])), turn_sigma=float(rs.choice([0.2, 0.6, 1.5])))
        if seed % 4 == 1:
            d["stroke_flags"] = capi.stroke_flags(cap, join, aa=False)
        got = run_async(rt, ctx_long, ps, d)
        assert got.status == 0
        assert_mesh_equal(got, pyoracle.tessellate(ps, d), "long %d" % seed)
        # (c) path-set tables
        ps = wl.fuzz_paths(seed, npaths=int(rs.randint(1, 90)), with_shapes=bool(seed & 1), degenerate=True) if seed % 3 else wl.thin_fuzz_paths(seed, npaths=int(rs.randint(1, 90)), degenerate=bool(seed & 2))
        pset = rt.PathSet(ctx_long, ps)
        desc = ps.desc()
        scal = None
        for which in (9, 0, 1, 2, 3, 4, 5, 6, 7, 8):
            n = C.c_uint64(0)
            rt._check(rt.lib().vgx_pathset_read_table(ctx_long.handle, pset.handle, which, None, 0, C.byref(n)), "read_table")
            a = np.zeros(max(int(n.value), 1), np.uint8)
            rt._check(rt.lib().vgx_pathset_read_table(ctx_long.handle, pset.handle, which, a.ctypes.data, a.nbytes, C.byref(n)), "read_table")
            hn = hostlib.vgxt_pathset_table(C.addressof(desc), which, None, 0)
            b = np.zeros(max(int(hn), 1), np.uint8)
            hostlib.vgxt_pathset_table(C.addressof(desc), which, b.ctypes.data, b.nbytes)
            a, b = a[:n.value], b[:hn]
            if which == 9:
                scal = a.view(np.uint32)
            if which == 6 and not scal[3]:
                a = a.view(np.uint32).reshape(-1, 4).copy(); b = b.view(np.uint32).reshape(-1, 4).copy()
                a[:, 0] &= 0xFFFF; b[:, 0] &= 0xFFFF; a[:, 3] = 0; b[:, 3] = 0
            assert np.array_equal(a, b), ("path-set table", which, seed)
        pset.close()
        # (d) vgx_tessellate's one-walk flatten route (k_flat1 + k_flatten_gather_ordered), every eligible batch: every draw its own path
        if seed % 4 == 0:
            ps = wl.fuzz_paths(seed, npaths=int(rs.randint(2100, 3200)), with_shapes=bool(seed & 4), degenerate=bool(seed & 8))
            d = wl.template_general_draws(ps, seed, 1, round_joins=True)
            got = run_async(rt, ctx_f1, ps, d, profile=True)
            assert got.status == 0 and "flatten_one_walk" in got.stages, got.stages
            assert_mesh_equal(got, pyoracle.tessellate(ps, d), "one-walk route %d" % seed)
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:300], flush=True)
print("round-6 soak done: %d seeds, mismatches: %d" % (N, bad))
