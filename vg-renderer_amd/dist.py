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


def shard_range_balanced(ctx, pset, draws_dev, ndraws, rank, world):
    """Contiguous draw range of `rank` when the batch is heterogeneous ("balance on the count-pass result"): every
    rank holds the whole batch's draw records, runs vgx_partition on them (deterministic: all ranks compute the same bounds) and
    takes its own range. Homogeneous batches give the ranges of shard_range."""
    from . import runtime as rt
    bounds, _ = rt.partition(ctx, pset, draws_dev, ndraws, world)
    return bounds[rank], bounds[rank + 1]


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


# ---- the same gather through the C-ABI (vgx_gather_sizes / vgx_gather, include/vgx.h) ------------------------------------
class CapiGather:
    """Runs libvgx's own RCCL gather from a torch.distributed job: creates a dedicated RCCL communicator on the copy of
    librccl the process already carries (torch's), one rank per process, and hands it to vgx_gather. This is what a C++
    host would do with its own communicator; torch is only used to broadcast the 128-byte unique id."""

    def __init__(self, ctx, device_index, group=None):
        import ctypes as C
        import os
        from . import capi, runtime
        self.C, self.capi, self.rt, self.ctx = C, capi, runtime, ctx
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        cand = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"]
        self.lib = None
        for c in cand:
            try:
                self.lib = C.CDLL(c)
                break
            except OSError:
                continue
        if self.lib is None:
            raise RuntimeError("no librccl found")

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid = UniqueId()
        if self.rank == 0:
            r = self.lib.ncclGetUniqueId(C.byref(uid))
            if r != 0:
                raise RuntimeError("ncclGetUniqueId -> %d" % r)
        backend = dist.get_backend(group)
        raw = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
        if backend == "nccl":
            raw = raw.to(torch.device("cuda", device_index))
        dist.broadcast(raw, 0, group=group)
        C.memmove(C.byref(uid), bytes(raw.cpu().numpy().tobytes()), 128)
        self.comm = C.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        torch.cuda.set_device(device_index)
        r = self.lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)
        if r != 0:
            raise RuntimeError("ncclCommInitRank -> %d" % r)

    def sizes(self, nverts, nidx, nmeshes, ndraws):
        C = self.C
        mine = self.capi.RankSizes(nverts, nidx, nmeshes, ndraws)
        allz = (self.capi.RankSizes * self.world)()
        self.rt._check(self.rt.lib().vgx_gather_sizes(self.ctx.handle, self.comm, C.byref(mine), allz, self.rt._stream_ptr()), "vgx_gather_sizes")
        return allz

    def gather(self, bufs, allz, root=0, global_bufs=None):
        """bufs / global_bufs: runtime.MeshBuffers. Enqueues on the current stream; no host synchronisation."""
        C = self.C
        local = bufs.out_struct()
        me = allz[self.rank]
        local.cap_vertices = max(local.cap_vertices, me.num_vertices)
        g = global_bufs.out_struct() if global_bufs is not None else None
        self.rt._check(self.rt.lib().vgx_gather(self.ctx.handle, self.comm, root, C.byref(local), allz, C.byref(g) if g is not None else None, self.rt._stream_ptr()), "vgx_gather")

    def gather_at(self, bufs, allz, place, root=0, global_bufs=None):
        """vgx_gather_at: rank r's block lands at place[r] (RankSizes used as offsets). For frames tessellated in tiles."""
        C = self.C
        local = bufs.out_struct()
        g = global_bufs.out_struct() if global_bufs is not None else None
        self.rt._check(self.rt.lib().vgx_gather_at(self.ctx.handle, self.comm, root, C.byref(local), allz, place, C.byref(g) if g is not None else None,
                                                   self.rt._stream_ptr()), "vgx_gather_at")

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy.argtypes = [self.C.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None
