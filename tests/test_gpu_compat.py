"""GPU: the C++ vg::pathXXX / vg::strokerXXX compat layer (include/vgx_compat.hpp, libvgx_compat.so) against the
CPU oracle, driven by one C++ program that issues the same call sequence to both (tests/compat_test.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vg-renderer_amd")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backend", ["device", "auto"])
def test_compat_layer_matches_oracle(oracle, backend):
    """device: every call is a vgx_* call sequence on the GPU; auto: the host's lane code, the GPU above VGX_COMPAT_DEVICE_MIN
    vertices (set low here so that both are exercised in one run). The host backend alone: tests/test_compat_context.py (no GPU)."""
    exe = os.path.join(ROOT, "tests", "compat_test.bin")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "compat_test.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"), "-I" + os.path.join(ROOT, "oracle", "bx_shim"),
                           "-I" + os.path.join(PKG, "csrc"), "-L" + PKG, "-lvgx_compat", "-lvgx", "-L" + os.path.join(ROOT, "oracle"), "-lvgoracle",
                           "-ldl", "-Wl,-rpath," + PKG, "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    env = dict(os.environ, VGX_COMPAT_BACKEND=backend, VGX_COMPAT_DEVICE_MIN="24")
    libref = os.path.join(ROOT, "oracle", "_ref", "libvgref.so")
    if os.path.exists(libref):  # concave fills: libtess2 + the reference's strokerConcaveFillEndAA live in there
        env["VGX_TEST_LIBVGREF"] = libref
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "OK:" in r.stdout
