"""CPU, world_size 2, gloo: the sharding + variable-size gather plumbing of vg-renderer_amd/dist.py.
The per-rank streams come from the CPU oracle here (no GPU in this container); what is under test is that the
gathered buffers are byte-identical to the unsharded result -- the property the 8-GPU run relies on."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, instances, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = importlib.import_module("vg-renderer_amd.workloads")
    dm = importlib.import_module("vg-renderer_amd.dist")
    import pyoracle
    lo, hi = dm.shard_range(instances, rank, world)
    ps, ops = wl.tiger_paths()
    draws = wl.tiger_draws(ops, hi - lo, first_instance=lo)
    r = pyoracle.tessellate(ps, draws)
    res = dm.gather_streams(torch.from_numpy(r.pos), torch.from_numpy(r.color.view(np.int32)), torch.from_numpy(r.idx.view(np.int16)),
                            torch.from_numpy(r.meshes.view(np.uint8).copy()), r.sizes["num_vertices"], r.sizes["num_indices"], r.sizes["num_meshes"],
                            draws.shape[0], root=0)
    if rank == 0:
        full = pyoracle.tessellate(ps, wl.tiger_draws(ops, instances))
        ok = (np.array_equal(res["pos"].numpy().view(np.uint32), full.pos.view(np.uint32))
              and np.array_equal(res["color"].numpy().view(np.uint32), full.color)
              and np.array_equal(res["idx"].numpy().view(np.uint16), full.idx)
              and np.array_equal(res["meshes_u8"].numpy()[:full.meshes.shape[0] * 32], full.meshes.view(np.uint8)))
        out_q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("instances", [3, 4])
def test_sharded_gather_equals_single_rank(instances):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + instances
    procs = [ctx.Process(target=_worker, args=(r, 2, port, instances, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_shard_range_partitions_exactly():
    dm = importlib.import_module("vg-renderer_amd.dist")
    for n in (0, 1, 7, 8, 80000):
        for w in (1, 2, 3, 8):
            r = [dm.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
