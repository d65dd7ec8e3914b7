"""Multi-GPU plumbing: contiguous sharding of draws over ranks and the variable-size gather of the
vertex / colour / index / mesh-table streams to a root rank over RCCL (torch.distributed "nccl" backend on
ROCm; "gloo" on CPU for the tests).

Independent path instances shard embarrassingly: rank r tessellates a contiguous range of
draws; mesh indices are mesh-local uint16, so concatenating the per-rank streams in rank order gives
byte-for-byte the single-GPU result -- only the mesh table's first_vertex / first_index / draw fields
need the rank's base added. There is exactly one exchange step:
  1. all_gather of the four per-rank totals,
  2. one isend/irecv per stream per peer (4 large messages per peer, each riding the peer's own xGMI
     link into the root), batched with batch_isend_irecv,
  3. a tiny rebase of the gathered mesh table on the root.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous [lo, hi) of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_streams(pos, color, idx, meshes_u8, nverts, nidx, nmeshes, ndraws_local, root=0, group=None):
    """Gather variable-length device streams to `root`.
    pos [nv,2] f32, color [nv] i32, idx [ni] i16, meshes_u8 [nm*32] u8 (vgx_mesh records).
    Returns on root: dict(pos, color, idx, meshes_u8, counts[world,4]); on other ranks: None."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = pos.device
    mine = torch.tensor([nverts, nidx, nmeshes, ndraws_local], dtype=torch.int64, device=dev)
    allc = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allc, mine, group=group)
    counts = torch.stack(allc).cpu().numpy()
    if rank != root:
        ops = []
        if nverts:
            ops.append(dist.P2POp(dist.isend, pos[:nverts].contiguous(), root, group))
            ops.append(dist.P2POp(dist.isend, color[:nverts].contiguous(), root, group))
        if nidx:
            ops.append(dist.P2POp(dist.isend, idx[:nidx].contiguous(), root, group))
        if nmeshes:
            ops.append(dist.P2POp(dist.isend, meshes_u8[:nmeshes * 32].contiguous(), root, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return None
    tv, ti, tm = int(counts[:, 0].sum()), int(counts[:, 1].sum()), int(counts[:, 2].sum())
    gpos = torch.empty((max(tv, 1), 2), dtype=pos.dtype, device=dev)
    gcol = torch.empty(max(tv, 1), dtype=color.dtype, device=dev)
    gidx = torch.empty(max(ti, 1), dtype=idx.dtype, device=dev)
    gm = torch.empty(max(tm, 1) * 32, dtype=torch.uint8, device=dev)
    vo = np.concatenate([[0], np.cumsum(counts[:, 0])])
    io = np.concatenate([[0], np.cumsum(counts[:, 1])])
    mo = np.concatenate([[0], np.cumsum(counts[:, 2])])
    do = np.concatenate([[0], np.cumsum(counts[:, 3])])
    ops = []
    for r in range(world):
        nv, ni, nm = int(counts[r, 0]), int(counts[r, 1]), int(counts[r, 2])
        if r == root:
            gpos[vo[r]:vo[r] + nv].copy_(pos[:nv])
            gcol[vo[r]:vo[r] + nv].copy_(color[:nv])
            gidx[io[r]:io[r] + ni].copy_(idx[:ni])
            gm[mo[r] * 32:(mo[r] + nm) * 32].copy_(meshes_u8[:nm * 32])
            continue
        if nv:
            ops.append(dist.P2POp(dist.irecv, gpos[vo[r]:vo[r] + nv], r, group))
            ops.append(dist.P2POp(dist.irecv, gcol[vo[r]:vo[r] + nv], r, group))
        if ni:
            ops.append(dist.P2POp(dist.irecv, gidx[io[r]:io[r] + ni], r, group))
        if nm:
            ops.append(dist.P2POp(dist.irecv, gm[mo[r] * 32:(mo[r] + nm) * 32], r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    # rebase the mesh table: first_vertex (u64 @0), first_index (u64 @8), draw (u32 @24) per 32-byte record
    if tm:
        rec64 = gm[:tm * 32].view(torch.int64).view(tm, 4)
        rec32 = gm[:tm * 32].view(torch.int32).view(tm, 8)
        for r in range(world):
            a, b = int(mo[r]), int(mo[r + 1])
            if b > a and r > 0:
                rec64[a:b, 0] += int(vo[r])
                rec64[a:b, 1] += int(io[r])
                rec32[a:b, 6] += int(do[r])
    return dict(pos=gpos, color=gcol, idx=gidx, meshes_u8=gm, counts=counts)
