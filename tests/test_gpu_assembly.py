"""Draw-command assembly on the GPU (vgx_set_assembly + the emit kernels' index base) against the oracle: vertex-buffer
partition, draw-command table and the rebased index buffer, bit-exact; vertex streams and mesh table unchanged."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    return importlib.import_module("vg-renderer_amd.runtime")


def run_assembled(rt, gpu_ctx, ps, d, max_vb, use_async):
    import torch
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    cap = 2 * (nv // (max_vb or 65536)) + 2
    cmds = torch.zeros(cap * 48, dtype=torch.uint8, device=dd.device)
    ncmd = torch.zeros(1, dtype=torch.int64, device=dd.device)
    gpu_ctx.set_assembly(cmds, max_vb, ncmd)
    try:
        if use_async:
            rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
        else:
            rt.tessellate_emit(gpu_ctx, pset, dd, d.shape[0], bufs)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_assembly(None)
    n = int(ncmd.item())
    out = dict(
        status=int(bufs.dev_status.item()) if use_async else 0,
        num=n,
        cmds=cmds[:n * 48].cpu().numpy().view(rt.capi.drawcmd_dtype),
        idx=bufs.idx[:ni].cpu().numpy().view(np.uint16),
        pos=bufs.pos[:nv].cpu().numpy(),
        color=bufs.color[:nv].cpu().numpy().view(np.uint32),
        meshes=bufs.meshes[:nm * 32].cpu().numpy().view(rt.capi.mesh_dtype),
        dev_sizes=bufs.dev_sizes.cpu().numpy() if use_async else None,
    )
    pset.close()
    return out


@pytest.mark.parametrize("max_vb", [0, 4096, 700])
@pytest.mark.parametrize("use_async", [False, True])
def test_assembly_tiger(rt, gpu_ctx, wl, oracle, max_vb, use_async):
    ps, d = wl.tiger(12)
    ref = oracle.tessellate(ps, d)
    st, rcmds, ridx = oracle.assemble(ref.meshes, ref.idx, max_vb)
    assert st == 0
    got = run_assembled(rt, gpu_ctx, ps, d, max_vb, use_async)
    assert got["status"] == 0
    assert got["num"] == len(rcmds), (got["num"], len(rcmds))
    for f in rcmds.dtype.names:
        assert np.array_equal(got["cmds"][f], rcmds[f]), f
    assert np.array_equal(got["idx"], ridx)
    assert np.array_equal(got["pos"].view(np.uint32), ref.pos.view(np.uint32))
    assert np.array_equal(got["color"], ref.color)
    for f in ref.meshes.dtype.names:
        assert np.array_equal(got["meshes"][f], ref.meshes[f]), f
    if use_async:
        assert int(got["dev_sizes"][9]) == len(rcmds)


@pytest.mark.parametrize("seed", [21, 22])
def test_assembly_fuzz_all_strokers(rt, gpu_ctx, wl, oracle, seed):
    """Every stroker kind (Round caps / joins write part of their indices directly, the rest through the register
    stage) with small vertex buffers, so that most meshes carry a non-zero base."""
    ps = wl.fuzz_paths(seed, npaths=96)
    d = wl.fuzz_draws(ps, seed)
    d = np.concatenate([d, d])
    ref = oracle.tessellate(ps, d)
    max_vb = int(max(2048, ref.meshes["num_vertices"].max()))
    st, rcmds, ridx = oracle.assemble(ref.meshes, ref.idx, max_vb)
    assert st == 0 and len(rcmds) > 3
    got = run_assembled(rt, gpu_ctx, ps, d, max_vb, True)
    assert got["status"] == 0 and got["num"] == len(rcmds)
    for f in rcmds.dtype.names:
        assert np.array_equal(got["cmds"][f], rcmds[f]), f
    assert np.array_equal(got["idx"], ridx)


def test_assembly_mesh_too_large_and_capacity(rt, gpu_ctx, wl, oracle):
    import torch
    ps, d = wl.tiger(2)
    ref = oracle.tessellate(ps, d)
    small = int(ref.meshes["num_vertices"].max()) - 1  # one mesh cannot fit any vertex buffer (vg.cpp:5323)
    got = run_assembled(rt, gpu_ctx, ps, d, small, True)
    assert got["status"] == rt.capi.VGX_E_MESH_TOO_LARGE
    # draw-command table too small: reported, never overrun
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    cmds = torch.full((2 * 48 + 48,), 0xAB, dtype=torch.uint8, device=dd.device)
    gpu_ctx.set_assembly(cmds[:96], 500)
    try:
        rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_assembly(None)
    assert int(bufs.dev_status.item()) == rt.capi.VGX_E_NOSPACE
    assert bool((cmds[96:] == 0xAB).all().item())
    pset.close()


@pytest.mark.parametrize("max_vb", [0, 2048, 700])
def test_state_key_split_and_uv_stream(rt, gpu_ctx, wl, oracle, max_vb):
    """VGX_ASM_SPLIT_STATE: draws carry a state key (DrawCommand type / handle / forced-new generation folded by the host);
    a key change between consecutive meshes starts a draw command inside the same vertex buffer (vg.cpp:5376-5379), the
    index rebase is relative to the COMMAND. Plus the white-pixel UV stream (vg.cpp:5218-5225), int16 x 2 and float x 2."""
    import torch
    ps, d = wl.tiger(3)
    rs = np.random.RandomState(11)
    # runs of 1..40 draws share a key; some runs return to an earlier key (merging must still not cross the other run)
    keys = np.repeat(rs.randint(1, 6, size=d.shape[0]).astype(np.uint32) << 16 | 3, rs.randint(1, 40, size=d.shape[0]))[:d.shape[0]]
    d = d.copy()
    d["state_key"] = keys
    ref = oracle.tessellate(ps, d)
    mesh_keys = d["state_key"][ref.meshes["draw"]]
    st, rcmds, ridx = oracle.assemble(ref.meshes, ref.idx, max_vb, mesh_keys=mesh_keys)
    assert st == 0
    pset = rt.PathSet(gpu_ctx, ps)
    dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(gpu_ctx, pset, dd, d.shape[0])
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    for uv_dtype, uv_value in ((torch.int16, (0x7FFF0001,)), (torch.float32, (0x3F000000, 0x3E800000))):
        bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
        cmds = torch.zeros((len(rcmds) + 4) * 48, dtype=torch.uint8, device=dd.device)
        ncmd = torch.zeros(1, dtype=torch.int64, device=dd.device)
        uv = torch.zeros((nv + 3, 2), dtype=uv_dtype, device=dd.device)
        gpu_ctx.set_assembly(cmds, max_vb, ncmd, split_state=True, uv=uv, uv_value=uv_value)
        try:
            rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
            torch.cuda.synchronize()
        finally:
            gpu_ctx.set_assembly(None)
        assert int(bufs.dev_status.item()) == 0
        n = int(ncmd.item())
        assert n == len(rcmds)
        got = cmds[:n * 48].cpu().numpy().view(rt.capi.drawcmd_dtype)
        for f in rt.capi.drawcmd_dtype.names:
            assert np.array_equal(got[f], rcmds[f]), f
        assert np.array_equal(bufs.idx[:ni].cpu().numpy().view(np.uint16), ridx)
        assert np.array_equal(bufs.pos[:nv].cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
        raw = uv.cpu().numpy().view(np.uint32).reshape(nv + 3, -1)
        assert (raw[:nv, 0] == uv_value[0]).all() and (raw[nv:] == 0).all()  # every vertex, nothing past the last one
        if uv_dtype == torch.float32:
            assert (raw[:nv, 1] == uv_value[1]).all()
    # a table one entry too small is reported, never overrun
    bufs = rt.MeshBuffers(dd.device, nv, ni, nm)
    cmds = torch.full((len(rcmds) * 48,), 0xAB, dtype=torch.uint8, device=dd.device)
    gpu_ctx.set_assembly(cmds[:(len(rcmds) - 1) * 48], max_vb, None, split_state=True)
    try:
        rt.tessellate_async(gpu_ctx, pset, dd, d.shape[0], bufs)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_assembly(None)
    assert int(bufs.dev_status.item()) == rt.capi.VGX_E_NOSPACE
    assert bool((cmds[(len(rcmds) - 1) * 48:] == 0xAB).all().item())
    pset.close()
