"""CPU: the frame-level rules around concave fills, pinned against the reference's own Context without a GPU --
vgx_cmdlist_decode's concave draws (VGX_FILL_CONCAVE: flags, colour, position in the draw sequence, the "a sub-path below
three vertices = no mesh at all" rule of ctxFillPath*, src/vg.cpp:3139-3141), the merge of their meshes into the frame by
draw index and the assembly of the merged mesh sequence. The device-side pieces are played by their oracles here (the
reference's path / stroker / libtess2 from oracle/_ref); tests/test_gpu_concave_frame.py runs the same frames through the
product (vgx_tessellate, vgx_flatten, vgx_concave_move / _emit, vgx_merge)."""
import importlib

import numpy as np
import pytest

import frameref as F
import concave_frame as CF
import test_gpu_concave as TC
import test_gpu_concave_frame as T


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.available("reference") or not F.R.available():
        pytest.skip("oracle/_ref is not built")
    return TC.load_ref(oracle)


def cpu_frame(oracle, ref, script, max_vb):
    rt = importlib.import_module("vg-renderer_amd.runtime")
    capi = rt.capi
    refd = F.reference_frame(script, max_vb=max_vb)
    ps, draws, n, extra = F.decode(rt, refd)
    assert n["skipped"] == 0
    A = oracle.tessellate(ps, draws)
    cidx = np.flatnonzero((draws["fill_flags"] & capi.FILL_CONCAVE) != 0)
    assert not (draws["fill_flags"][cidx] & 1).any()  # a concave draw never has VGX_FILL_ENABLE
    B = []
    if cidx.shape[0]:
        fl = oracle.flatten(ps, draws[cidx], apply_transform=True)
        for k, di in enumerate(cidx):
            info = fl.draw_info[k]
            subs = fl.subpaths[int(info["first_subpath"]):int(info["first_subpath"]) + int(info["num_subpaths"])]
            if subs.shape[0] == 0 or (subs["num_vertices"] < 3).any():
                continue
            contours = [fl.poly[int(s["first_vertex"]):int(s["first_vertex"]) + int(s["num_vertices"])] for s in subs]
            ff = int(draws["fill_flags"][di])
            eo = 1 if ff & capi.FILL_EVEN_ODD else 0
            col = int(draws["fill_color"][di])
            if ff & capi.FILL_AA:
                pos, c, idx = TC._reference_mesh(ref, contours, col, float(draws["fringe"][di]), eo)
            else:
                pos, idx = CF._polygons(ref, contours, eo)
                c = np.full(pos.shape[0], col, np.uint32)
            B.append((int(di), pos, c, idx))
    seq = []
    for m in A.meshes:
        v0, nv, i0, ni = int(m["first_vertex"]), int(m["num_vertices"]), int(m["first_index"]), int(m["num_indices"])
        seq.append((int(m["draw"]), 0, A.pos[v0:v0 + nv], A.color[v0:v0 + nv], A.idx[i0:i0 + ni], int(m["subpath_kind"])))
    for di, pos, c, idx in B:
        seq.append((di, 1, pos, c, idx, capi.MESH_CONCAVE_FILL_AA << 28))
    seq.sort(key=lambda t: (t[0], t[1]))
    meshes = np.zeros(len(seq), dtype=capi.mesh_dtype)
    v = i = 0
    for k, t in enumerate(seq):
        meshes[k] = (v, i, t[2].shape[0], t[4].shape[0], t[0], t[5])
        v += t[2].shape[0]
        i += t[4].shape[0]
    pos = np.concatenate([t[2] for t in seq])
    col = np.concatenate([t[3] for t in seq])
    idx = np.concatenate([t[4] for t in seq])
    st, cmds, idx2 = oracle.assemble(meshes, idx, max_vb, mesh_keys=draws["state_key"][meshes["draw"]])
    assert st == 0
    F.assert_frame_equal(refd["frame"], pos, col, idx2, meshes, cmds, draws, extra["draw_state"], max_vb)
    return len(B)


@pytest.mark.parametrize("max_vb", [65536, 2048])
def test_concave_scenario_cpu(oracle, ref, max_vb):
    assert cpu_frame(oracle, ref, T.s_concave(), max_vb) == 7


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_concave_frames_cpu(oracle, ref, seed):
    cpu_frame(oracle, ref, T.s_random_concave(4000 + seed), 65536 if seed % 2 else 4096)
