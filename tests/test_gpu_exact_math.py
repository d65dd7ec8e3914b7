"""csrc/vgx_fastmath.h (correctly rounded 1/x, sqrt, 1/sqrt in a few instructions, used by the element kernels) against
the compiler's correctly rounded `/` and sqrtf over EVERY binary32 value of the functions' domains (5 billion values,
about a second on an MI355X). tests/native/exact_math_test.hip is the program; __graft_entry__.build() compiles it."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fastmath_is_correctly_rounded_over_its_whole_domain():
    exe = os.path.join(ROOT, "tests", "native", "exact_math_test.bin")
    if not os.path.exists(exe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-o", exe,
                               os.path.join(ROOT, "tests", "native", "exact_math_test.hip")])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if "mismatches=" in l]
    assert r.returncode == 0 and len(lines) == 3, r.stdout
    for l in lines:
        assert " mismatches=0 of " in l, l
