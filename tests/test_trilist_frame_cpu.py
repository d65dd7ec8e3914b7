"""CPU: vg::indexedTriList in a frame, pinned against the reference's own Context without a GPU -- vgx_cmdlist_decode's tri_*
arrays (positions through the state transform, colours replicated, UVs copied or the white pixel's, the draw's image / scissor /
place in the sequence) merged into the frame by draw index and assembled into draw commands. The device-side pieces are played
by their oracles here; tests/test_gpu_trilist_frame.py runs the same frames through the product (vgx_tessellate, vgx_merge_uv)."""
import importlib

import numpy as np
import pytest

import frameref as F
import trilist_frame as TF
import test_concave_frame_cpu as CC
import test_gpu_concave as TC
import concave_frame as CF


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.available("reference") or not F.R.available():
        pytest.skip("oracle/_ref is not built")
    return TC.load_ref(oracle)


def compose(oracle, ref, refd, ps, draws, extra, max_vb):
    """Sequence A (oracle tessellation) + concave fills (reference stroker + libtess2) + user meshes -> merged by draw, assembled."""
    rt = importlib.import_module("vg-renderer_amd.runtime")
    capi = rt.capi
    A = oracle.tessellate(ps, draws)
    seq = []
    for m in A.meshes:
        v0, nv, i0, ni = int(m["first_vertex"]), int(m["num_vertices"]), int(m["first_index"]), int(m["num_indices"])
        seq.append((int(m["draw"]), 0, A.pos[v0:v0 + nv], A.color[v0:v0 + nv], A.idx[i0:i0 + ni], int(m["subpath_kind"]), None))
    cidx = np.flatnonzero((draws["fill_flags"] & capi.FILL_CONCAVE) != 0)
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
            seq.append((int(di), 1, pos, c, idx, capi.MESH_CONCAVE_FILL_AA << 28, None))
    tri = extra["tri"]
    for m in tri["meshes"]:
        v0, nv, i0, ni = int(m["first_vertex"]), int(m["num_vertices"]), int(m["first_index"]), int(m["num_indices"])
        assert int(m["subpath_kind"]) == capi.MESH_TRILIST << 28
        assert int(draws["fill_flags"][int(m["draw"])]) == capi.FILL_TRILIST
        seq.append((int(m["draw"]), 2, tri["pos"][v0:v0 + nv], tri["color"][v0:v0 + nv], tri["idx"][i0:i0 + ni], int(m["subpath_kind"]), tri["uv"][v0:v0 + nv]))
    seq.sort(key=lambda t: (t[0], t[1]))
    meshes = np.zeros(len(seq), dtype=capi.mesh_dtype)
    v = i = 0
    for k, t in enumerate(seq):
        meshes[k] = (v, i, t[2].shape[0], t[4].shape[0], t[0], t[5])
        v += t[2].shape[0]
        i += t[4].shape[0]
    pos = np.concatenate([t[2] for t in seq]) if seq else np.zeros((0, 2), np.float32)
    col = np.concatenate([t[3] for t in seq]) if seq else np.zeros(0, np.uint32)
    idx = np.concatenate([t[4] for t in seq]) if seq else np.zeros(0, np.uint16)
    white, nb = refd["white_uv"]
    uv = np.zeros((pos.shape[0], 2), tri["uv"].dtype)
    uv[:] = np.frombuffer(white.tobytes()[:nb], dtype=tri["uv"].dtype)
    for k, t in enumerate(seq):
        if t[6] is not None:
            uv[int(meshes["first_vertex"][k]):int(meshes["first_vertex"][k]) + t[6].shape[0]] = t[6]
    st, cmds, idx2 = oracle.assemble(meshes, idx, max_vb, mesh_keys=draws["state_key"][meshes["draw"]])
    assert st == 0
    return pos, col, idx2, meshes, cmds, uv


def cpu_frame(oracle, ref, script, max_vb, uv_float=False):
    rt = importlib.import_module("vg-renderer_amd.runtime")
    refd = F.reference_frame(script, max_vb=max_vb, uv_float=uv_float, images=6)
    ps, draws, n, extra = F.decode(rt, refd)
    assert n["skipped"] == 0
    pos, col, idx, meshes, cmds, uv = compose(oracle, ref, refd, ps, draws, extra, max_vb)
    F.assert_frame_equal(refd["frame"], pos, col, idx, meshes, cmds, draws, extra["draw_state"], max_vb, uv=uv)
    return extra["tri"]["meshes"].shape[0]


@pytest.mark.parametrize("uv_float", [False, True])
@pytest.mark.parametrize("max_vb", [65536, 512])
def test_trilist_scenario_cpu(oracle, ref, max_vb, uv_float):
    assert cpu_frame(oracle, ref, TF.s_trilist(uv_float), max_vb, uv_float) == 8


@pytest.mark.parametrize("uv_float", [False, True])
def test_trilist_only_cpu(oracle, ref, uv_float):
    assert cpu_frame(oracle, ref, TF.s_trilist_only(uv_float), 65536, uv_float) == 4


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_trilist_frames_cpu(oracle, ref, seed):
    cpu_frame(oracle, ref, TF.s_random(seed, bool(seed & 1)), 65536 if seed % 3 else 1024, bool(seed & 1))


def test_trilist_store_pass_needs_the_arrays(oracle):
    """A list with user meshes decoded by a caller that hands over no tri_* arrays: VGX_E_NOSPACE, not a frame with holes."""
    import ctypes as C
    rt = importlib.import_module("vg-renderer_amd.runtime")
    capi = rt.capi
    refd = F.reference_frame(TF.s_trilist_only(), images=6)
    data = refd["bytes"]
    st = capi.CmdListState()
    st.mtx[0] = st.mtx[3] = 1.0
    st.global_alpha = 1.0; st.tess_tol = 0.25; st.fringe = 1.0; st.canvas_width, st.canvas_height = 1280.0, 720.0
    out = capi.CmdListOut()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    assert rt.lib().vgx_cmdlist_decode(buf, len(data), C.byref(st), C.byref(out)) == 0
    assert (out.num_tri_meshes, out.num_draws) == (4, 4) and out.num_paths == 1 and out.num_tri_vertices > 0
    draws = np.zeros(4, capi.draw_dtype)
    pcb = np.zeros(2, np.uint32)
    ct, ao, ar = np.zeros(1, np.uint8), np.zeros(1, np.uint32), np.zeros(1, np.float32)
    out.cmd_type, out.cmd_arg_off, out.args, out.path_cmd_begin, out.draws = ct.ctypes.data, ao.ctypes.data, ar.ctypes.data, pcb.ctypes.data, draws.ctypes.data
    out.cap_cmds, out.cap_args, out.cap_paths, out.cap_draws = 0, 0, 1, 4
    assert rt.lib().vgx_cmdlist_decode(buf, len(data), C.byref(st), C.byref(out)) == capi.VGX_E_NOSPACE
