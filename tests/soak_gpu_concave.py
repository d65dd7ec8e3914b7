"""GPU soak of the concave-fill path (not collected by pytest): N random fill sets (tests/test_gpu_concave.py::_random_fills) through
the reference's strokerConcaveFillEndAA and through vgx_concave_move / _emit. `python tests/soak_gpu_concave.py 300`."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pyoracle
import test_gpu_concave as T
rt = importlib.import_module("vg-renderer_amd.runtime")
ctx = rt.Context(0)
ref = T.load_ref(pyoracle)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for seed in range(5000, 5000 + n):
    try:
        T._check_fills(rt, ctx, ref, T._random_fills(seed))
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:200])
print("concave soak done: %d seeds, mismatches: %d" % (n, bad))
