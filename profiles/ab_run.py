"""Tuning helper: same-box A/B timing of experimental libvgx builds (profiles/ab_variants.sh). Box-to-box and run-to-run
variation of the memory-bound kernels is +-15 %, so the variants are run round-robin ROUNDS times, each in a fresh
process (VGX_LIB is read at import), and the per-stage MINIMUM and median over the rounds are printed.
  python profiles/ab_run.py [--rounds 3] [--stages fill_emit,stroke_emit] head run4 ...   (names of vg-renderer_amd/dbg/libvgx_<name>.so)"""
import ast
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = 3
stages = None
which = "tiger"
while args and args[0].startswith("--"):
    if args[0] == "--rounds":
        rounds = int(args[1])
    elif args[0] == "--stages":
        stages = args[1].split(",")
    elif args[0] == "--workload":
        which = args[1]
    args = args[2:]
acc = {v: {} for v in args}
for r in range(rounds):
    for v in args:
        env = dict(os.environ)
        if v != "default":
            env["VGX_LIB"] = os.path.join(ROOT, "vg-renderer_amd", "dbg", "libvgx_%s.so" % v)
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "stage_times.py"), which], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120).stdout.strip().splitlines()[-1]
            d = ast.literal_eval(out[out.index("{"):])
        except (subprocess.TimeoutExpired, IndexError, ValueError, SyntaxError):
            print("variant %s: run failed (round %d)" % (v, r), flush=True)
            continue
        d["total"] = sum(d.values())
        for k, x in d.items():
            acc[v].setdefault(k, []).append(x)
keys = stages or ["total", "flatten_build", "fill_emit", "stroke_emit"]
print("%-14s" % "variant" + "".join("%26s" % (k + " min/med") for k in keys))
for v in args:
    if not acc[v]:
        print("%-14s no successful run" % v)
        continue
    print("%-14s" % v + "".join("%26s" % ("%.3f / %.3f" % (min(acc[v][k]), statistics.median(acc[v][k]))) for k in keys))
