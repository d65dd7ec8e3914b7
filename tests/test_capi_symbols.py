"""CPU: the C-ABI library loads, exports every symbol include/vgx.h declares, validates path grammar on the
host, and fails loudly (no CPU fallback) when there is no device."""
import importlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rt():
    m = importlib.import_module("vg-renderer_amd.runtime")
    if not os.path.exists(m.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return m


def test_every_header_symbol_is_exported(rt, vgr):
    hdr = open(os.path.join(ROOT, "include", "vgx.h")).read()
    declared = set(re.findall(r"\b(vgx_[a-z_]+)\s*\(", hdr))
    declared -= {"vgx_cmd", "vgx_status"}
    assert declared == set(vgr.capi.VGX_SYMBOLS.keys()), declared ^ set(vgr.capi.VGX_SYMBOLS.keys())
    lib = rt.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.vgx_version() == 1
    assert lib.vgx_status_string(4) == b"VGX_E_NOSPACE"


def test_struct_layouts_match_header(vgr):
    import ctypes as C
    capi = vgr.capi
    assert capi.draw_dtype.itemsize == 64 and capi.mesh_dtype.itemsize == 32
    assert capi.subpath_dtype.itemsize == 16 and capi.draw_info_dtype.itemsize == 40
    assert C.sizeof(capi.Sizes) == 80 and C.sizeof(capi.Assembly) == 56 and C.sizeof(capi.MeshOut) == 56 and C.sizeof(capi.FlatOut) == 40
    assert C.sizeof(capi.PathSetDesc) == 40


def test_path_grammar_validation(rt, vgr):
    capi = vgr.capi

    def status(build):
        b = vgr.PathSetBuilder()
        b.begin_path()
        build(b)
        b.end_path()
        return rt.validate_pathset(b.arrays())

    assert status(lambda b: (b.move_to(0, 0), b.line_to(1, 1), b.close())) == capi.VGX_OK
    assert status(lambda b: (b.rect(0, 0, 1, 1), b.circle(0, 0, 2), b.move_to(0, 0), b.cubic_to(1, 1, 2, 2, 3, 3))) == capi.VGX_OK
    assert status(lambda b: (b.arc(0, 0, 5, 0, 1, True), b.line_to(3, 3))) == capi.VGX_OK
    # lineTo before moveTo: only VG_CHECKed in the reference's debug build (path.cpp:82)
    assert status(lambda b: b.line_to(1, 1)) == capi.VGX_E_INVALID_PATH
    # adding to a closed sub-path (path.cpp:765)
    assert status(lambda b: (b.move_to(0, 0), b.line_to(1, 0), b.line_to(1, 1), b.close(), b.line_to(2, 2))) == capi.VGX_E_INVALID_PATH
    assert status(lambda b: (b.rect(0, 0, 1, 1), b.line_to(2, 2))) == capi.VGX_E_INVALID_PATH
    # NaN would hang the reference's subdivision loop (path.cpp:109)
    assert status(lambda b: (b.move_to(0, 0), b.cubic_to(float("nan"), 0, 1, 1, 2, 2))) == capi.VGX_E_NONFINITE
    assert status(lambda b: (b.move_to(0, 0), b.line_to(float("inf"), 0))) == capi.VGX_E_NONFINITE


def test_no_cpu_fallback_without_device(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    import ctypes as C
    h = C.c_void_p()
    assert rt.lib().vgx_create(0, C.byref(h)) == 7  # VGX_E_NO_DEVICE
    with pytest.raises(RuntimeError):
        rt.Context(0)


def test_product_code_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under vg-renderer_amd/ or include/ may reference it."""
    bad = []
    for base in ("vg-renderer_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"pyoracle|libvgoracle|libvgref|vgo_tessellate|vgo_flatten|vgo_port|ref_capi", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_cpp_hosts_compile_and_link_against_the_c_abi(rt, tmp_path):
    """The C++ programs that drive the C-ABI (the example, the multi-rank gather test with its stand-in RCCL, the
    reference-API compat test) compile and link here, without a GPU: catches header / symbol drift on the CPU round."""
    import subprocess
    pkg = os.path.join(ROOT, "vg-renderer_amd")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-I", inc, os.path.join(ROOT, "examples", "vgx_example.cpp"),
                           "-L", pkg, "-lvgx", "-Wl,-rpath," + pkg, "-o", str(tmp_path / "vgx_example")])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-I", inc, os.path.join(ROOT, "tests", "native", "gather_test.cpp"),
                           "-L", pkg, "-lvgx", "-L/opt/rocm/lib", "-lrccl", "-ldl", "-lpthread", "-Wl,-rpath," + pkg, "-o", str(tmp_path / "gather_test")])
    subprocess.check_call(["g++", "-shared", "-fPIC", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "native", "fake_rccl.cpp"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-o", str(tmp_path / "libfake_rccl.so")])
    # the stand-in exports exactly the entry points vgx_gather binds
    syms = subprocess.check_output(["nm", "-D", "--defined-only", str(tmp_path / "libfake_rccl.so")], text=True)
    for name in ("ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclAllGather", "ncclCommCount", "ncclCommUserRank"):
        assert (" T " + name) in syms, name
