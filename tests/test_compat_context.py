"""The drop-in check of SURVEY 8(b), on the CPU: the reference's own Context (src/vg.cpp compiled unmodified) linked against the
PRODUCT's per-call API -- libvgx_compat.so: vg::pathXXX / vg::strokerXXX of include/vgx_compat.hpp served by the product's lane
code on the host (vg-renderer_amd/host/vgx_host_backend.hip), libtess2 handed over by the application -- instead of against its
own path.cpp / stroker.cpp. Every frame scenario of the suite is played on both Contexts (oracle/_ref/libvgref_vg_compat.so and
oracle/_ref/libvgref_vg.so) and what vg::end() hands to bgfx is compared byte for byte: vertex buffers (positions, UVs, colours),
the index buffer, the draw and clip command tables. No GPU is involved: the per-call boundary runs on the host.
tests/compat_test.cpp compares the same API call by call (both backends) with the oracle."""
import os
import subprocess
import time

import numpy as np
import pytest

import pyvgref as R
import frameref as F
import test_cmdlist_ref as TR
import trilist_frame as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not (R.available() and os.path.exists(R.PATH_COMPAT)), reason="oracle/_ref/libvgref_vg[_compat].so not built (needs /root/reference)")


@pytest.fixture(autouse=True)
def host_backend(monkeypatch):
    monkeypatch.setenv("VGX_COMPAT_BACKEND", "host")


def same_frame(a, b, what):
    fa, fb = a["frame"], b["frame"]
    assert len(fa.vbs) == len(fb.vbs), what
    for va, vb in zip(fa.vbs, fb.vbs):
        for k in ("pos", "uv", "color"):
            assert va[k].shape == vb[k].shape and va[k].tobytes() == vb[k].tobytes(), (what, k)
    assert fa.idx.tobytes() == fb.idx.tobytes(), (what, "idx")
    assert fa.drawcmds.tobytes() == fb.drawcmds.tobytes(), (what, "draw commands")
    assert fa.clipcmds.tobytes() == fb.clipcmds.tobytes(), (what, "clip commands")
    assert fa.submits.tobytes() == fb.submits.tobytes(), (what, "bgfx submits")
    return sum(v["pos"].shape[0] for v in fa.vbs)


def both(script, what, **kw):
    ref = F.reference_frame(script, **kw)
    got = F.reference_frame(script, compat=True, **kw)
    return same_frame(ref, got, what)


@pytest.mark.parametrize("name", ["tiger", "paints", "scissor_clip", "latch", "every_command"])
@pytest.mark.parametrize("immediate", [False, True])
def test_scenarios_render_identically_over_the_compat_library(wl, name, immediate):
    script = getattr(TR, "s_" + name)(wl)
    assert both(script, name, immediate=immediate, max_vb=65536) > 0


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_frames_render_identically_over_the_compat_library(seed):
    assert both(TR.s_random(7000 + seed), "random %d" % seed, max_vb=65536 if seed % 3 else 2048) > 0


def test_concave_and_user_mesh_frames_render_identically_over_the_compat_library():
    import test_gpu_concave_frame as TC
    assert both(TC.s_concave(), "concave", max_vb=65536) > 0
    for seed in range(6):
        both(TC.s_random_concave(4000 + seed), "random concave %d" % seed, max_vb=65536)
    assert both(TF.s_trilist(), "trilist", images=6) > 0


def test_cached_command_list_renders_identically_over_the_compat_library(wl):
    script = TR.s_cached_drawing(wl)
    for frames in (1, 2):  # second frame: from the shape cache (clCacheRender)
        both(script, "cached x%d" % frames, flags=R.CL_CACHEABLE, frames=frames)


def test_tiger_frame_time_is_reference_class(wl):
    """The per-call boundary is the host's job (~1 us per call): a Tiger-like frame through the reference's Context costs about
    the same over libvgx_compat.so as over the reference's own path.cpp / stroker.cpp."""
    script = TR.s_tiger(wl, K=20)
    t = {}
    for compat in (False, True):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            F.reference_frame(script, immediate=True, compat=compat)
            best = min(best, time.perf_counter() - t0)
        t[compat] = best
    print("tiger x20 frame (python-driven calls included): reference %.1f ms, over libvgx_compat %.1f ms" % (t[False] * 1e3, t[True] * 1e3))
    assert t[True] < 3.0 * t[False] + 0.05


def test_compat_api_matches_oracle_on_the_host_backend(oracle):
    """tests/compat_test.cpp (every vg::pathXXX / strokerXXX call against the oracle, concave fills against the reference's own
    strokerConcaveFillEndAA) with the host backend: no GPU."""
    pkg = os.path.join(ROOT, "vg-renderer_amd")
    exe = os.path.join(ROOT, "tests", "compat_test_host.bin")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "compat_test.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"), "-I" + os.path.join(ROOT, "oracle", "bx_shim"),
                           "-I" + os.path.join(pkg, "csrc"), "-L" + pkg, "-lvgx_compat", "-lvgx", "-L" + os.path.join(ROOT, "oracle"), "-lvgoracle",
                           "-ldl", "-Wl,-rpath," + pkg, "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    env = dict(os.environ, VGX_COMPAT_BACKEND="host")
    libref = os.path.join(ROOT, "oracle", "_ref", "libvgref.so")
    if os.path.exists(libref):
        env["VGX_TEST_LIBVGREF"] = libref
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "OK:" in r.stdout, r.stdout[-2000:]
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=dict(env, VGX_COMPAT_TIMING="1"))
    print(r.stdout)  # one drawing = 14 API calls: a few microseconds on the host backend
