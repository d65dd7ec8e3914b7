"""GPU soak of whole frames (not collected by pytest): N random frames (tests/test_cmdlist_ref.py::s_random, recorded with the
reference's own writers) through vgx_cmdlist_decode -> vgx_tessellate with assembly armed, against what the reference's own
Context hands to bgfx. `python tests/soak_gpu_frames.py 500 [tight]` (tight: vertex buffers sized just above the largest mesh)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import frameref as F
import test_cmdlist_ref as T
rt = importlib.import_module("vg-renderer_amd.runtime"); wl = importlib.import_module("vg-renderer_amd.workloads")
ctx = rt.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tight = len(sys.argv) > 2 and sys.argv[2] == "tight"
bad = skipped = 0
for seed in range(300000, 300000 + n):
    script = T.s_random(seed)
    max_vb = 65536 if seed % 2 else 8192
    try:
        ref = F.reference_frame(script, max_vb=max_vb)
        ps, draws, nn, extra = F.decode(rt, ref)
        if len(ref["frame"].drawcmds) == 0:
            skipped += 1
            continue
        if tight:  # vertex buffers just above the frame's largest mesh: a buffer switch every few meshes
            import pyoracle
            max_vb = int(pyoracle.tessellate(ps, draws).meshes["num_vertices"].max()) + seed % 7
            ref = F.reference_frame(script, max_vb=max_vb)
        white, nb = ref["white_uv"]
        got = T.gpu_frame(rt, ctx, ps, draws, max_vb, uv_bytes=nb, uv_value=int(white[0]))
        F.assert_frame_equal(ref["frame"], got["pos"], got["color"], got["idx"], got["meshes"], got["cmds"], draws, extra["draw_state"], max_vb,
                             uv=got["uv"], what="gpu random %d" % seed)
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:200])
print("gpu frame soak done: %d seeds, %d without draw commands, mismatches: %d" % (n, skipped, bad))
