"""vgx_gather_sizes / vgx_gather (the RCCL gather behind the C-ABI) driven from C++ (tests/native/gather_test.cpp):
  - 1 rank over the REAL librccl (binding by dlopen, ncclAllGather, local copies, capacity check),
  - 2 and 3 ranks as threads on the one GPU over tests/native/fake_rccl.cpp (RCCL refuses two ranks on one device): shard ->
    tessellate -> gather -> byte-identical streams and mesh table to a single context tessellating the whole batch,
    root = first and last rank (so the root's own block is rebased too).
The real multi-GPU transfer is only exercised by the driver's N-GPU bench (bench.py --gpus N)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vg-renderer_amd")


@pytest.fixture(scope="module")
def binaries(tmp_path_factory):
    d = tmp_path_factory.mktemp("gather")
    exe, fake = str(d / "gather_test"), str(d / "libfake_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "gather_test.cpp"),
                           "-L", PKG, "-lvgx", "-L/opt/rocm/lib", "-lrccl", "-ldl", "-lpthread", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    subprocess.check_call(["g++", "-shared", "-fPIC", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "native", "fake_rccl.cpp"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", fake])
    return exe, fake


def test_gather_one_rank_real_rccl(binaries):
    exe, _ = binaries
    out = subprocess.check_output([exe, "real"], text=True, timeout=300)
    assert "ranks 1 root 0" in out and "identical to the single-context run" in out


@pytest.mark.parametrize("nranks,ndraws", [(2, 1000), (3, 1000), (4, 3), (4, 7)])
def test_gather_threaded_ranks(binaries, nranks, ndraws):
    """(4, 3): fewer draws than ranks -- empty shards send and receive nothing and still take part in the size exchange."""
    exe, fake = binaries
    out = subprocess.check_output([exe, "fake", fake, str(nranks), str(ndraws)], text=True, timeout=300)
    assert out.count("identical to the single-context run") == 2, out
    assert "MISMATCH" not in out


@pytest.mark.parametrize("nranks,ndraws,tiles", [(2, 4000, 3), (3, 2000, 2), (4, 5, 2)])
def test_gather_at_tiled_frames(binaries, nranks, ndraws, tiles):
    """vgx_gather_at: every rank cuts its draws into tiles, tessellates them one after the other and gathers tile by tile
    (what lets the gather of tile t overlap the tessellation of tile t + 1); with VGX_GATHER_CHUNK_MB=1 every transfer
    leaves in several pieces. The gathered frame must be byte-identical to a single context tessellating everything."""
    exe, fake = binaries
    env = dict(os.environ, VGX_GATHER_CHUNK_MB="1")
    out = subprocess.check_output([exe, "tiles", fake, str(nranks), str(ndraws), str(tiles)], text=True, timeout=300, env=env)
    assert out.count("identical to the single-context run") == 2, out
    assert "MISMATCH" not in out


def test_gather_eight_ranks_into_a_70_gb_destination(binaries):
    """The root of BASELINE config 4 (Tiger x80k over 8 GPUs) receives 8 x 8.75 GB. Eight thread-ranks on the one GPU send the
    same 8.75 GB local block (one tessellation, shared) into one 70 GB destination: 64-bit offsets, capacity checks and the
    piece splitting (default 256 MiB) at full size; first / last megabyte of every block and the rebased mesh records checked."""
    import torch
    free, total = torch.cuda.mem_get_info(0)
    if free < 110 * (1 << 30):
        pytest.skip("needs ~95 GB of free device memory")
    exe, fake = binaries
    out = subprocess.check_output([exe, "big", fake, "8", "8.75"], text=True, timeout=900)
    assert "blocks identical, mesh records rebased" in out and "MISMATCH" not in out, out


def _expected_partition(oracle, ps, d, nparts):
    """The bounds vgx_partition must produce, from the reference's own per-draw polyline counts (include/vgx.h)."""
    import importlib
    capi = importlib.import_module("vg-renderer_amd.capi")
    di = oracle.flatten(ps, d).draw_info
    ff, sf = d["fill_flags"].astype(np.uint64), d["stroke_flags"].astype(np.uint64)
    f = np.where(ff & 1, np.where(ff & 2, 2, 1), 0) + np.where(sf & 1, np.where((sf & capi.STROKE_AA) == 0, 2, np.where(sf & capi.STROKE_THIN, 3, 4)), 0)
    w = di["num_poly_vertices"].astype(np.uint64) * f.astype(np.uint64) + 1
    prefix = np.concatenate([[0], np.cumsum(w)]).astype(np.uint64)
    total = int(prefix[-1])
    bounds = [int(np.searchsorted(prefix[:-1], (total * k) // nparts, side="left")) for k in range(nparts)] + [d.shape[0]]
    weights = [int(prefix[bounds[k + 1]] - prefix[bounds[k]]) for k in range(nparts)]
    return bounds, weights


@pytest.mark.parametrize("nparts", [1, 3, 8])
def test_partition_balances_a_heterogeneous_batch(wl, oracle, nparts):
    """vgx_partition ("for heterogeneous batches balance on the count-pass result"): a batch whose draws differ 100x in
    size, sorted so that equal draw counts per rank would be badly unbalanced. Bounds = the cut points of the predicted-output
    prefix (checked against the reference's per-draw counts); every part within one draw's weight of total / nparts; tessellating
    the parts separately and concatenating them in part order reproduces the whole batch byte for byte."""
    import importlib
    import torch
    rt = importlib.import_module("vg-renderer_amd.runtime")
    ps = wl.fuzz_paths(910, npaths=64, with_shapes=True, with_polylines=True)
    base = wl.fuzz_draws(ps, 910)
    d = np.concatenate([base] * 6)
    ref_counts = oracle.flatten(ps, d).draw_info["num_poly_vertices"]
    d = d[np.argsort(ref_counts, kind="stable")]  # small draws first: equal counts per rank = unequal work
    ctx = rt.Context(0)
    try:
        pset = rt.PathSet(ctx, ps)
        dd = rt.upload_draws(d)
        bounds, weights = rt.partition(ctx, pset, dd, d.shape[0], nparts)
        eb, ew = _expected_partition(oracle, ps, d, nparts)
        assert bounds == eb and weights == ew
        assert bounds[0] == 0 and bounds[-1] == d.shape[0] and all(a <= b for a, b in zip(bounds, bounds[1:]))
        total = sum(weights)
        biggest = int(_per_draw_weight(oracle, ps, d).max())
        assert max(weights) - total / nparts <= biggest, (weights, total / nparts, biggest)
        if nparts == 3:
            equal = [sum(ew2) for ew2 in np.array_split(_per_draw_weight(oracle, ps, d), nparts)]
            assert max(equal) > 1.5 * max(weights)  # what the naive split would have cost the slowest rank
        # parts tessellated separately == the whole batch (mesh-local indices: no rebase of the streams)
        whole = rt.tessellate(ctx, pset, dd, d.shape[0])
        pos, idx, col = [], [], []
        for k in range(nparts):
            lo, hi = bounds[k], bounds[k + 1]
            if lo == hi:
                continue
            part = rt.tessellate(ctx, pset, rt.upload_draws(d[lo:hi]), hi - lo)
            pos.append(part.pos); idx.append(part.idx); col.append(part.color)
        assert np.array_equal(np.concatenate(pos).view(np.uint32), whole.pos.view(np.uint32))
        assert np.array_equal(np.concatenate(idx), whole.idx) and np.array_equal(np.concatenate(col), whole.color)
        pset.close()
    finally:
        ctx.close()
        torch.cuda.synchronize()


def _per_draw_weight(oracle, ps, d):
    import importlib
    capi = importlib.import_module("vg-renderer_amd.capi")
    di = oracle.flatten(ps, d).draw_info
    ff, sf = d["fill_flags"].astype(np.uint64), d["stroke_flags"].astype(np.uint64)
    f = np.where(ff & 1, np.where(ff & 2, 2, 1), 0) + np.where(sf & 1, np.where((sf & capi.STROKE_AA) == 0, 2, np.where(sf & capi.STROKE_THIN, 3, 4)), 0)
    return (di["num_poly_vertices"].astype(np.uint64) * f.astype(np.uint64) + 1).astype(np.int64)


def test_partition_edge_cases(wl, oracle):
    """More parts than draws (empty parts at the front of equal bounds), a single draw, one part."""
    import importlib
    rt = importlib.import_module("vg-renderer_amd.runtime")
    ps = wl.fuzz_paths(911, npaths=8, with_shapes=False, with_polylines=True)
    base = wl.fuzz_draws(ps, 911)
    ctx = rt.Context(0)
    try:
        pset = rt.PathSet(ctx, ps)
        for d, nparts in ((base[:3], 8), (base[:1], 4), (base, 1), (base[:5], 5)):
            dd = rt.upload_draws(d)
            bounds, weights = rt.partition(ctx, pset, dd, d.shape[0], nparts)
            eb, ew = _expected_partition(oracle, ps, d, nparts)
            assert bounds == eb and weights == ew, (bounds, eb)
            assert bounds[0] == 0 and bounds[-1] == d.shape[0] and all(a <= b for a, b in zip(bounds, bounds[1:]))
            assert sum(weights) == int(_per_draw_weight(oracle, ps, d).sum())
        pset.close()
    finally:
        ctx.close()
