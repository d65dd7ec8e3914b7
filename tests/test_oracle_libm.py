"""The pinned float32 transcendentals (csrc/vgmath.h) against glibc's, in ULPs.

Kernels, restatement and the compiled reference all take cos / sin / tan / acos / atan2 / rsqrt from vgmath.h (bx, the
reference's math library, is neither vendored nor version-pinned), so a wrong polynomial there would be invisible to
every bit-exact parity test. This file bounds the distance of each function from an independent implementation on the
ranges the path uses (angles of arcs and round joins, dot products in [-1, 1], squared lengths)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "vg-renderer_amd", "libvgx_hosttest.so")


@pytest.fixture(scope="module")
def hostlib():
    src = os.path.join(ROOT, "vg-renderer_amd", "csrc", "vgx_hosttest.cpp")
    import glob
    newest = max(os.path.getmtime(f) for f in [src] + glob.glob(os.path.join(os.path.dirname(src), "*.h")))  # (the lane code lives in the headers)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-shared", "-o", LIB, src])
    lib = C.CDLL(LIB)
    lib.vgxt_math_vec.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    return lib


def _ulps(got, want64):
    """Distance of float32 `got` from the float64 reference value, in units of the float32 spacing at the reference."""
    want32 = want64.astype(np.float32)
    spacing = np.spacing(np.abs(want32)).astype(np.float64)
    spacing = np.maximum(spacing, np.float64(np.finfo(np.float32).tiny))
    return np.abs(got.astype(np.float64) - want64) / spacing


def _run(lib, fn, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b if b is not None else a, dtype=np.float32)
    out = np.empty_like(a)
    lib.vgxt_math_vec(fn, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.shape[0])
    return out


def test_sin_cos_tan_close_to_libm(hostlib):
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.uniform(-4 * np.pi, 4 * np.pi, 200000), rs.uniform(-1e-3, 1e-3, 20000), np.linspace(-100.0, 100.0, 50001)]).astype(np.float32)
    x64 = x.astype(np.float64)
    # absolute error relative to 1 ulp of 1.0 near the zeros (|result| small), relative elsewhere
    for fn, ref in ((0, np.cos), (1, np.sin)):
        got = _run(hostlib, fn, x)
        err = np.abs(got.astype(np.float64) - ref(x64))
        assert float(err.max()) <= 4 * 2.0 ** -24, (fn, float(err.max()))
    # tan only feeds pathArcTo's tangent length (path.cpp:203-300) at half-angles away from the poles
    xt = rs.uniform(-1.4, 1.4, 100000).astype(np.float32)
    got = _run(hostlib, 2, xt)
    assert float(_ulps(got, np.tan(xt.astype(np.float64))).max()) <= 8.0


def test_acos_atan2_rsqrt_close_to_libm(hostlib):
    rs = np.random.RandomState(1)
    x = np.concatenate([rs.uniform(-1.0, 1.0, 200000), 1.0 - np.logspace(-8, 0, 2000), -1.0 + np.logspace(-8, 0, 2000), [1.0, -1.0, 0.0]]).astype(np.float32)
    got = _run(hostlib, 3, x)
    err = np.abs(got.astype(np.float64) - np.arccos(x.astype(np.float64)))
    assert float(err.max()) <= 4e-6, float(err.max())  # acos near +-1 is ill-conditioned: absolute bound (angles are then divided by da >= 1e-2)
    # outside [-1, 1] by a rounding error (dot products of unit vectors): clamped, never NaN -- the reference's arc loops
    # would not terminate on NaN (oracle/libm_sensitivity.py shows the glibc build hanging on exactly that)
    edge = _run(hostlib, 3, np.array([1.0000001, -1.0000001, 1.5, -1.5], dtype=np.float32))
    assert np.isfinite(edge).all()
    y = rs.uniform(-1000.0, 1000.0, 200000).astype(np.float32)
    z = rs.uniform(-1000.0, 1000.0, 200000).astype(np.float32)
    got = _run(hostlib, 4, y, z)
    err = np.abs(got.astype(np.float64) - np.arctan2(y.astype(np.float64), z.astype(np.float64)))
    assert float(err.max()) <= 1e-6, float(err.max())
    s = np.exp(rs.uniform(np.log(1e-12), np.log(1e12), 200000)).astype(np.float32)
    got = _run(hostlib, 5, s)
    assert float(_ulps(got, 1.0 / np.sqrt(s.astype(np.float64))).max()) <= 1.5
