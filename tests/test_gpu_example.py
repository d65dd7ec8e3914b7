"""The C-ABI used directly from C++ (examples/vgx_example.cpp, no Python / torch in the process): builds, runs, and
reports the config-0 known answer (68 vertices / 300 indices per stroked cubic) for every instance."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_example_runs(tmp_path):
    exe = str(tmp_path / "vgx_example")
    pkg = os.path.join(ROOT, "vg-renderer_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "vgx_example.cpp"),
                           "-L", pkg, "-lvgx", "-Wl,-rpath," + pkg, "-o", exe])
    out = subprocess.check_output([exe, "1000"], text=True)
    assert "instances 1000  meshes 1000  vertices 68000  indices 300000  polyline vertices 17000" in out
    assert "mesh 0: 68 vertices, 300 indices" in out


def test_cpp_frame_example_runs(tmp_path):
    """examples/vgx_frame_example.cpp: command-list bytes (paths + IndexedTriList user meshes) -> vgx_cmdlist_decode -> vgx_tessellate + vgx_merge_uv
    with assembly armed, all from C++."""
    exe = str(tmp_path / "vgx_frame_example")
    pkg = os.path.join(ROOT, "vg-renderer_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "vgx_frame_example.cpp"),
                           "-L", pkg, "-lvgx", "-Wl,-rpath," + pkg, "-o", exe])
    out = subprocess.check_output([exe], text=True)
    assert "300 paths" in out and "453 draws (3 of them user meshes), 0 skipped" in out and "consistent" in out and "INCONSISTENT" not in out, out


def test_cpp_static_scene_example_runs(tmp_path):
    """examples/vgx_static_scene.cpp: a retained scene (400 distinct paths, Round-join strokes) under a moving camera as a static batch
    (vgx_set_static_batches), per-frame sizes that follow the camera, VGX_E_STALE on a structural change and the recount that answers it."""
    exe = str(tmp_path / "vgx_static_scene")
    pkg = os.path.join(ROOT, "vg-renderer_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "vgx_static_scene.cpp"),
                           "-L", pkg, "-lvgx", "-Wl,-rpath," + pkg, "-o", exe])
    out = subprocess.check_output([exe, "30"], text=True)
    assert "30 frames as a static batch" in out and "after swapping two draws: VGX_E_STALE" in out and "after counting again: VGX_OK" in out, out
