import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vgr():
    return importlib.import_module("vg-renderer_amd")


@pytest.fixture(scope="session")
def wl():
    return importlib.import_module("vg-renderer_amd.workloads")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle loader (test infrastructure). Builds oracle/libvgoracle.so on demand."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "libvgoracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libvgoracle.so"])
    import pyoracle
    return pyoracle


@pytest.fixture(scope="session")
def gpu_ctx():
    rt = importlib.import_module("vg-renderer_amd.runtime")
    ctx = rt.Context(0)
    yield ctx
    ctx.close()
