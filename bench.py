#!/usr/bin/env python
"""bench.py -- headline benchmark: M tessellated verts/sec (stroke + fill AA), Tiger x10k batch.

One "step" = one pass of the whole hot path over one batch that is already resident in HBM:
  path commands + draw records (HBM)  ->  flatten (count, scan, emit)  ->  transformPath
  ->  strokerConvexFillAA / strokerPolylineStrokeAA[Thin] per sub-path (count, scan, emit)
  ->  vertex / colour / index / mesh-table streams (HBM)
through the asynchronous C-ABI entry point vgx_tessellate (no host round trip inside the step).

Launch contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver runs it under
torch.distributed.run with one rank per GPU. Every rank tessellates its own contiguous range of Tiger
instances (independent path instances shard embarrassingly, no data-path collective) => weak scaling;
`value` = vertices produced by ALL ranks per second of the slowest rank. With more than one rank the RCCL gather of
the streams to rank 0 (vg-renderer_amd/dist.py; SURVEY 8e counts it into the scaling target) is timed too, by default,
and reported as `gather_ms` / `value_with_gather` beside `value`.

Rank 0 prints ONE JSON line (fields documented in DESIGN.md "Measurement"): the headline (Tiger x10k = BASELINE's
metric) with `roofline` and `cpu_baseline`, and -- on 1-GPU runs -- the other two single-GPU BASELINE configs (1M
cubics flatten-only, 10k x 1k-segment Round/Round polylines) under `configs`, each with its own roofline object.
`--config NAME` makes another config the headline of the line.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0  # G wave64 VALU instructions / s: 1024 SIMDs, one instruction per 2 cycles, 2.4 GHz (MI355X_MICROARCH.md)


def command_bytes(ps, draw_paths):
    """Bytes of path commands the batch reads: 1 B opcode + 4 B argument offset + 4 B per argument, per command instance."""
    import numpy as np
    pcb = ps.path_cmd_begin.astype(np.int64)
    aoff = ps.cmd_arg_off.astype(np.int64)
    per_path = (pcb[1:] - pcb[:-1]) * 5 + (aoff[pcb[1:]] - aoff[pcb[:-1]]) * 4
    return int(per_path[draw_paths].sum())


def algorithmic_bytes(cmd_bytes, ndraws, sizes, fill_verts, fill_idx, fill_meshes, instanced=False, template=False):
    """Algorithmic HBM bytes per launch of each kernel (stated in DESIGN.md):
    commands read once per instance (1 B opcode + 4 B arg offset + 4 B per argument), one 64 B draw record
    per path instance, 8 B per polyline vertex, 12 B per output vertex (float2 position + uint32 colour), 2 B per
    index, 32 B per mesh record. Scratch the kernels exchange (per-command words, mesh descriptors, prefix arrays,
    scan partials) is NOT counted: it is overhead, not algorithm."""
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    ef = sizes["num_fill_elements"]
    es = sizes["num_elements"] - ef
    b = {}
    # the instanced kernel (one lane per instance, 64 instances of a path per wave) reads a path's commands once per wave,
    # not once per instance; both write one 16-byte record per sub-path besides the vertices
    b["flatten_build"] = (cmd_bytes // 64 if instanced else cmd_bytes) + 64 * ndraws + 8 * sizes["num_poly_vertices"] + 16 * sizes["num_subpaths"]
    b["flatten_count"] = cmd_bytes + 64 * ndraws
    b["flatten_emit"] = cmd_bytes + 64 * ndraws + 8 * sizes["num_poly_vertices"] + 16 * sizes["num_subpaths"]
    b["flatten_one_walk"] = b["flatten_emit"]  # vgx_flatten: the same bytes, read and written once
    b["fill_emit"] = 8 * ef + 12 * fill_verts + 2 * fill_idx + 32 * fill_meshes
    b["stroke_emit"] = 8 * es + 12 * (nv - fill_verts) + 2 * (ni - fill_idx) + 32 * (nm - fill_meshes)
    # round 6: batches of fills + closed Miter AA / Thin strokes are emitted by ONE draw-ordered tile kernel (k_emit_tiles): both kinds' bytes
    b["tile_emit"] = 8 * (ef + es) + 12 * nv + 2 * ni + 32 * nm
    # the whole step: commands are read once per 64-instance task by the instanced flatten kernel, once per instance otherwise
    b["pipeline"] = (cmd_bytes // 64 if instanced else cmd_bytes) + 64 * ndraws + 12 * nv + 2 * ni + 32 * nm
    if template:
        # template mode (vgx_tmpl.hip): the path commands are not read at all during a step (the first period was flattened by
        # vgx_tessellate_count; its local polyline, a few hundred KB, stays in L2) -- a step reads the draw records and writes
        # the output streams. k_tmpl_verify re-reads the draw records: overhead, counted as its own line only.
        b["tmpl_verify"] = 64 * ndraws
        b["tmpl_emit"] = 64 * ndraws + 12 * nv + 2 * ni + 32 * nm
        b["pipeline"] = b["tmpl_emit"]
    return b


def effective_cores():
    """Host cores this process may actually use: affinity mask clipped by the cgroup CPU quota (the GPU boxes
    report 256 logical CPUs but run the container under a quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(budget_seconds=10.0, max_procs=256, which="tiger"):
    """Reference CPU path on the host cores of THIS box: one process per usable core, every process loops over its own shard
    (tiger: 16 instances; cubics: 20 000 cubics; round: 8 polylines x 1000 segments) for `budget_seconds` of wall time after a
    common start. Returns the dict for the JSON line (value = sum of units / slowest process's wall)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    kind = "reference" if pyoracle.available("reference") else "port"
    if kind == "port" and not pyoracle.available("port"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libvgoracle.so"], stdout=subprocess.DEVNULL)
    procs = max(1, min(effective_cores(), max_procs))
    worker = os.path.join(ROOT, "oracle", "cpu_bench.py")
    shard = {"tiger": 16, "cubics": 20000, "round": 8, "tigerspec": 4, "varied": 21, "tigeropen": 16, "tigerbevel": 16, "tigerround": 16}[which]
    what = {"tiger": "Tiger x16 instances", "cubics": "20 000 independent cubics (flatten + transform only)", "round": "8 polylines x 1000 segments, Round joins + caps",
            "tigerspec": "the SURVEY-spec drawing x4 instances", "varied": "Tiger x21 instances at 7 scales under rotations",
            "tigeropen": "Tiger with open sub-paths x16 instances", "tigerbevel": "Tiger with Bevel joins x16 instances", "tigerround": "Tiger with Round joins x16 instances"}[which]
    unit = "M polyline verts/s" if which == "cubics" else "M verts/s"

    def run(k, nproc, budget):
        ps = [subprocess.Popen([sys.executable, worker, k, str(shard), str(i * shard), str(budget), which], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
              for i in range(nproc)]
        for p in ps:
            assert p.stdout.readline().strip() == "ready"
        for p in ps:
            p.stdin.write("go\n")
            p.stdin.flush()
        outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
        return sum(o["verts"] for o in outs), max(o["seconds"] for o in outs), sum(o["cpu_seconds"] for o in outs)

    v1, t1, _ = run(kind, 1, min(2.0, budget_seconds))
    vN, tN, cN = run(kind, procs, budget_seconds)
    sse = None
    if which == "tiger" and kind == "reference" and pyoracle.available("reference_sse"):
        # the reference's fastest configuration (SSE2 strokerConvexFillAA, stroker.cpp:368-711): speed only, its
        # indices / rounding differ from the scalar build, so it is not a parity oracle
        vs, ts, _ = run("reference_sse", procs, budget_seconds / 2)
        sse = round(vs / ts / 1e6, 2)
    return {"value": round(vN / tN / 1e6, 2), "unit": unit, "cores": procs, "kind": kind,
            "value_sse_stroker": sse,
            "single_core_value": round(v1 / t1 / 1e6, 2), "cpu_seconds_per_wall_second": round(cN / tN, 1),
            "sample": "%s per process, looped for %.0f s of wall time, %d processes (one per usable host core; "
                      "%d logical CPUs visible)" % (what, budget_seconds, procs, os.cpu_count() or 0)}


def gpu_environment():
    """Clock / power-management state of GPU 0 as rocm-smi reports it (the run-to-run spread of the memory-bound kernels follows
    it); best effort, never fails the bench."""
    env = {}
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showperflevel", "--showclocks", "--showcomputepartition", "--showmemorypartition", "--json"],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20).stdout
        j = json.loads(out[out.index("{"):])
        card = j.get("card0", {})
        for k, v in card.items():
            kl = k.lower()
            if "performance level" in kl:
                env["perf_level"] = v
            elif kl.startswith("sclk clock speed"):
                env["sclk"] = v  # the clock at the moment of the query (idle between runs: the low state)
            elif kl.startswith("mclk clock speed"):
                env["mclk"] = v
            elif kl.startswith("fclk clock speed"):
                env["fclk"] = v
            elif "compute partition" in kl:
                env["compute_partition"] = v
            elif "memory partition" in kl:
                env["memory_partition"] = v
    except Exception as e:  # noqa: BLE001
        env["error"] = repr(e)[:80]
    return env


WORKLOADS = {
    "tiger10k": "BASELINE configs[2]: Tiger x10k, convexFillAA + polylineStrokeAA (the headline)",
    "cubics1m": "BASELINE configs[1]: 1M independent cubics, adaptive flatten only",
    "round10k": "BASELINE configs[3]: 10k polylines x 1k segments, Round joins + Round caps",
    "tiger10k_varied": "Tiger x10k at 7 scales (0.5 .. 3.5; 18 distinct avgScale values after rounding) under rotations: template mode with one template per class",
    "tiger10k_open": "Tiger x10k with every sub-path left open (no pathClose): open Miter strokes with Butt caps, template mode's general kernel",
    "tiger10k_bevel": "Tiger x10k with Bevel joins on the strokes: template mode, closed Bevel routine beside the closed Miter one (k_tmpl_emit_bevel; round 4: the general element body)",
    "tiger10k_round": "Tiger x10k with Round joins on the strokes (arc points counted on every instance's transformed polyline): template mode with per-step sizes (k_tmpl_round_sizes + k_tmpl_emit_round_aa)",
    "tiger10k_round_ordinary": "the tiger10k_round batch with VGX_TMPL_ROUND=0: the ordinary pipeline (k_flatten_inst + k_fill + k_round_sizes + k_stroke), what Round joins cost before round 5",
    "tiger10k_varied_round": "the tiger10k_varied batch (7 scales, 18 classes) with Round joins on the strokes: class-aware Round-join templates (round 6) -- one template per class, per-step sizes counted per instance",
    "tiger10k_varied_round_ordinary": "the same batch with VGX_TMPL_ROUND=0: the ordinary pipeline (k_flatten_inst with the instances sorted by tolerance class + k_fill + k_round_sizes + k_stroke), what it cost before",
    "cubics1m_stroked": "BASELINE configs[1]'s million cubics STROKED (2 px, Butt / Miter, AA) through vgx_tessellate: unrelated draws with long curves -- the one-walk flatten route (k_flat1 under vgx_tessellate, round 6)",
    "cubics1m_stroked_heap_route": "the same batch with VGX_TESS_FLAT1=0: k_flatten_build (one lane per path command, leaves through a heap), what it cost before",
    "round10k_static": "BASELINE configs[3]'s batch (10k polylines x 1k segments, Round joins + Round caps) with vgx_set_static_batches: the draw list flattened once by the count, a step = per-step Round-join sizes + the template emit kernel (no flatten, no scans)",
    "tiger10k_culled": "Tiger x10k after culling and reordering (a random 70 % of the draws, shuffled: no period left) with vgx_set_static_batches: the draw list flattened once by the count as ONE template, a step = the emit kernel (element tables from HBM instead of L2)",
    "tiger10k_culled_ordinary": "the tiger10k_culled batch without static batches: k_flatten_inst grouped by path + k_fill + k_stroke, what such a scene cost before round 5",
    "tigerspec10k": "SURVEY 8(d) config 3 as specified: 240 paths x (1-4 sub-paths x 8-60 cubics), x10k instances",
    "tiger10k_animated": "Tiger x10k whose path ARGUMENTS change every step: new path set (validated + uploaded) -> vgx_tessellate_count (template rebuilt: the flattener runs) -> vgx_tessellate, all inside the timed region",
    # honesty configs: the headline batch WITHOUT the template mode (every instance flattened, polyline through HBM), and without
    # any instancing shortcut (what a batch of 2.4 M unrelated draws costs)
    "tiger10k_per_instance_flatten": "Tiger x10k with VGX_TMPL=0: k_flatten_inst (one lane per instance) + k_fill + k_stroke, the round-3 pipeline",
    "tiger10k_command_parallel": "Tiger x10k with VGX_INST=0: k_flatten_build (one lane per path command) + k_fill + k_stroke, no instancing at all",
    "tiger10k_varied_per_instance_flatten": "the tiger10k_varied batch with VGX_TMPL_CLASSES=0: what instances cost when they share no subdivision (k_flatten_inst with the instances sorted by tolerance class + k_fill + k_stroke)",
}
CONFIG_ENV = {"round10k_static": {"VGX_TMPL_BATCH": "1"}, "tiger10k_culled": {"VGX_TMPL_BATCH": "1"}, "tiger10k_culled_ordinary": {"VGX_TMPL_BATCH": "0"}, "tiger10k_round_ordinary": {"VGX_TMPL_ROUND": "0"}, "tiger10k_varied_round_ordinary": {"VGX_TMPL_ROUND": "0"}, "cubics1m_stroked_heap_route": {"VGX_TESS_FLAT1": "0"}, "tiger10k_per_instance_flatten": {"VGX_TMPL": "0"}, "tiger10k_command_parallel": {"VGX_INST": "0"}, "tiger10k_varied_per_instance_flatten": {"VGX_TMPL_CLASSES": "0"}}
# configs that get their own cpu_baseline (the honesty configs are the headline's batch: they share its baseline)
CONFIG_CPU = {"cubics1m": "cubics", "round10k": "round", "tigerspec10k": "tigerspec", "tiger10k_varied": "varied", "tiger10k_open": "tigeropen", "tiger10k_bevel": "tigerbevel", "tiger10k_round": "tigerround"}
CONFIG_CPU_BUDGET = {"cubics": 4.0, "round": 4.0}  # seconds of wall time per config (the tiger variants: 2.5 s)


def make_workload(wl, name, instances, rank):
    """(path set, draw records, description, kind) of one BASELINE config; kind 'tessellate' or 'flatten'."""
    if name in ("tiger10k", "tiger10k_per_instance_flatten", "tiger10k_command_parallel"):
        ps, ops = wl.tiger_paths()
        d = wl.tiger_draws(ops, instances, first_instance=rank * instances)
        return ps, d, ("tiger-like 240-path drawing (seed 2024) x %d instances per GPU: convexFillAA on every sub-path + "
                       "polylineStrokeAA/AAThin (Butt/Miter) on 1/3 of the paths" % instances), "tessellate"
    if name in ("tiger10k_varied", "tiger10k_varied_per_instance_flatten"):
        ps, ops = wl.tiger_paths()
        d = wl.tiger_varied_draws(ops, instances, first_instance=rank * instances)
        return ps, d, ("tiger-like drawing (seed 2024) x %d instances per GPU, every instance at its own scale in {0.5 .. 3.5} and "
                       "rotation (tolerance and stroke widths follow the scale)" % instances), "tessellate"
    if name in ("tiger10k_varied_round", "tiger10k_varied_round_ordinary"):
        ps, ops = wl.tiger_paths()
        d = wl.tiger_varied_draws(ops, instances, first_instance=rank * instances, join=1)  # vg::LineJoin::Round
        return ps, d, ("tiger-like drawing (seed 2024) x %d instances per GPU, every instance at its own scale in {0.5 .. 3.5} and rotation, "
                       "Round joins on the strokes" % instances), "tessellate"
    if name == "tiger10k_open":
        ps, ops = wl.tiger_paths(closed=False)
        d = wl.tiger_draws(ops, instances, first_instance=rank * instances)
        return ps, d, ("the tiger-like drawing (seed 2024) with open sub-paths x %d instances per GPU: convexFillAA + polylineStrokeAA/AAThin "
                       "with Butt caps and Miter joins on 1/3 of the paths" % instances), "tessellate"
    if name == "tiger10k_bevel":
        ps, ops = wl.tiger_paths()
        d = wl.tiger_draws(ops, instances, first_instance=rank * instances, join=2)  # vg::LineJoin::Bevel
        return ps, d, ("the tiger-like drawing (seed 2024) x %d instances per GPU: convexFillAA + polylineStrokeAA/AAThin with Bevel joins "
                       "on 1/3 of the paths" % instances), "tessellate"
    if name in ("tiger10k_round", "tiger10k_round_ordinary"):
        ps, ops = wl.tiger_paths()
        d = wl.tiger_draws(ops, instances, first_instance=rank * instances, join=1)  # vg::LineJoin::Round
        return ps, d, ("the tiger-like drawing (seed 2024) x %d instances per GPU: convexFillAA + polylineStrokeAA (Round joins) / AAThin "
                       "on 1/3 of the paths" % instances), "tessellate"
    if name in ("tiger10k_culled", "tiger10k_culled_ordinary"):
        ps, ops = wl.tiger_paths()
        d = wl.tiger_draws(ops, instances, first_instance=rank * instances)
        import numpy as np
        rs = np.random.RandomState(4711 + rank)
        d = d[rs.uniform(size=d.shape[0]) < 0.7]
        d = d[rs.permutation(d.shape[0])]
        return ps, d, ("the tiger-like drawing (seed 2024) x %d instances per GPU, a random 70 %% of the draws kept and shuffled (seed 4711): a culled, "
                       "reordered instanced scene without a period" % instances), "tessellate"
    if name == "tigerspec10k":
        ps, ops = wl.tiger_spec_paths()
        d = wl.tiger_draws(ops, instances, first_instance=rank * instances)
        return ps, d, ("SURVEY 8(d) drawing (seed 2025: 240 paths, 1-4 closed sub-paths of 8-60 cubics, 1/3 stroked with widths 0.5-3) "
                       "x %d instances per GPU" % instances), "tessellate"
    if name == "cubics1m":
        ps, d = wl.random_cubics(1000000, seed=1234 + rank, box=1000.0)
        return ps, d, "1 000 000 independent paths (moveTo + cubicTo, 8 coordinates uniform in [0,1000)) per GPU: pathXXX only (vgx_flatten_count + vgx_flatten_emit)", "flatten"
    if name in ("cubics1m_stroked", "cubics1m_stroked_heap_route"):
        ps, d = wl.random_cubics(1000000, seed=1234 + rank, box=1000.0)
        wl.set_stroke(d, slice(None), 0xFF2060A0, 2.0, wl.capi.CAP_BUTT, wl.capi.JOIN_MITER, aa=True)
        return ps, d, "1 000 000 independent paths (moveTo + cubicTo, coordinates uniform in [0,1000)) per GPU, each stroked 2 px wide (Butt caps, Miter joins, AA)", "tessellate"
    if name in ("round10k", "round10k_static"):
        ps, d = wl.random_walk_polylines(10000, 1000, seed=5678 + rank)
        return ps, d, "10 000 open polylines x 1000 segments per GPU, strokerPolylineStrokeAA with Round joins + Round caps, width 6", "tessellate"
    raise SystemExit("unknown --config %s (one of %s)" % (name, ", ".join(WORKLOADS)))


def hetero_leg(rt, torch, wl, dev, local_rank, rank, world, red_dev, instances=4000, steps=5):
    import numpy as np
    parts = world if world > 1 else 8
    info, err = {}, None
    bal, nai = [(float("nan"), 0)], [(float("nan"), 0)]
    try:  # local work only: the collective below runs whatever happens here, so that one failing rank cannot hang the others
        ps, ops = wl.tiger_paths()
        d = wl.tiger_varied_draws(ops, instances, first_instance=0)
        npth = len(ops)
        inst_scale = d["scale"].reshape(instances, npth)[:, 0]
        order = np.argsort(inst_scale, kind="stable")  # whole instances, small scales (coarse subdivisions, few vertices) first
        d = d.reshape(instances, npth)[order].reshape(-1)
        n = int(d.shape[0])
        ctx = rt.Context(local_rank)  # its own context: the headline's scratch and template stay as they are
        pset = rt.PathSet(ctx, ps)
        dd = rt.upload_draws(d, local_rank)
        bounds, weights = rt.partition(ctx, pset, dd, n, parts)
        naive = [n * k // parts for k in range(parts + 1)]

        def time_range(lo, hi):
            if hi <= lo:
                return 0.0, 0
            sl = dd[lo * 64:hi * 64]
            sz = rt.tessellate_count(ctx, pset, sl, hi - lo)
            bufs = rt.MeshBuffers(dev, sz["num_vertices"], sz["num_indices"], sz["num_meshes"])
            for _ in range(2):
                rt.tessellate_async(ctx, pset, sl, hi - lo, bufs)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                rt.tessellate_async(ctx, pset, sl, hi - lo, bufs)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / steps * 1e3
            assert int(bufs.dev_status.item()) == 0
            return ms, int(sz["num_vertices"])

        mine = [rank] if world > 1 else list(range(parts))
        bal = [time_range(bounds[k], bounds[k + 1]) for k in mine]
        nai = [time_range(naive[k], naive[k + 1]) for k in mine]
        pset.close()
        ctx.close()
        info = {"workload": "Tiger x%d at 7 scales (0.5 .. 3.5), instances sorted by scale (%d draws)" % (instances, n), "parts": parts,
                "how": "one rank per part" if world > 1 else "parts timed one after the other on one GPU",
                "vgx_partition_bounds": [int(b) for b in bounds], "predicted_weight_max_over_min": round(max(weights) / max(1, min(weights)), 4)}
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([bal[0][0], float(bal[0][1]), nai[0][0], float(nai[0][1])], dtype=torch.float64, device=red_dev)
        each = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(each, t)
        bal = [(float(e[0]), int(e[1])) for e in each]
        nai = [(float(e[2]), int(e[3])) for e in each]
    if err is not None:
        return {"error": err}
    bm, nm = [round(x[0], 3) for x in bal], [round(x[0], 3) for x in nai]
    info.update({"balanced_ms": bm, "balanced_verts": [x[1] for x in bal], "balanced_max_over_min": round(max(bm) / max(1e-9, min(bm)), 3),
                 "equal_count_ms": nm, "equal_count_verts": [x[1] for x in nai], "equal_count_max_over_min": round(max(nm) / max(1e-9, min(nm)), 3),
                 "slowest_rank_gain": round(max(nm) / max(1e-9, max(bm)), 3)})
    return info


def run_config(rt, torch, ctx, local_rank, name, ps, draws, kind, steps, warmup, barrier, placements=1):
    """Times `steps` steps of one workload (inputs resident in HBM) between barriers. Returns a dict with the wall time,
    the output sizes, the per-kernel HIP-event times (averaged over a few extra steps outside the timed region) and the
    algorithmic bytes per kernel."""
    import numpy as np
    dev = torch.device("cuda", local_rank)
    ndraws = draws.shape[0]
    cmd_bytes = command_bytes(ps, draws["path"])
    # "H2D of commands excluded and reported separately" -- what happens once per batch, before any step, timed here:
    # vgx_pathset_create (grammar validation + derived tables on the host + the upload of the path set), the upload of the draw
    # records, and the first count call (sizes the library's scratch: includes its hipMalloc calls).
    # Round 6: vgx_pathset_create uploads the caller's four arrays as they are (ordinary host memory) and builds every derived
    # table on the device; the draw records are written by the caller into pinned memory (rt.pin_draws, untimed: it stands for the
    # caller producing them) and h2d_draws is the copy from there.
    pinned = rt.pin_draws(draws)
    torch.cuda.synchronize()
    ts0 = time.perf_counter()
    pset = rt.PathSet(ctx, ps)
    ts1 = time.perf_counter()
    dd = rt.upload_draws(pinned, local_rank)
    torch.cuda.synchronize()
    ts2 = time.perf_counter()
    raw_bytes = int(ps.cmd_type.nbytes + ps.cmd_arg_off.nbytes + ps.args.nbytes + ps.path_cmd_begin.nbytes)
    res = {"ndraws": ndraws, "setup_ms": {"pathset_create": round((ts1 - ts0) * 1e3, 3), "h2d_draws": round((ts2 - ts1) * 1e3, 3),
                                          "h2d_draws_GBps": round(ndraws * 64 / max(ts2 - ts1, 1e-9) / 1e9, 1),
                                          "path_set_bytes": raw_bytes, "draw_bytes": int(ndraws) * 64}}
    stage_sum = {}
    L, C, capi = rt.lib(), rt.C, rt.capi
    if kind == "flatten":
        sizes_c = capi.Sizes()
        tc0 = time.perf_counter()
        rt._check(L.vgx_flatten_count(ctx.handle, pset.handle, dd.data_ptr(), ndraws, C.byref(sizes_c), rt._stream_ptr()), "vgx_flatten_count")
        res["setup_ms"]["first_count"] = round((time.perf_counter() - tc0) * 1e3, 3)
        sizes = sizes_c.as_dict()
        npv, nsp = sizes["num_poly_vertices"], sizes["num_subpaths"]
        poly = torch.empty((max(npv, 1), 2), dtype=torch.float32, device=dev)
        subs = torch.empty(max(nsp, 1) * 16, dtype=torch.uint8, device=dev)
        dinfo = torch.empty(max(ndraws, 1) * 40, dtype=torch.uint8, device=dev)
        out = capi.FlatOut(poly.data_ptr(), subs.data_ptr(), dinfo.data_ptr(), npv, nsp)

        fb = rt.FlatBuffers(dev, npv, nsp, ndraws)  # vgx_flatten: the caller's buffers, exact capacities
        two_phase = os.environ.get("VGX_BENCH_FLATTEN") == "two_phase"
        res["flatten_entry"] = "vgx_flatten_count + vgx_flatten_emit (two walks, host round trips)" if two_phase else "vgx_flatten (one walk, asynchronous)"

        def step2(collect=None):
            rt._check(L.vgx_flatten_count(ctx.handle, pset.handle, dd.data_ptr(), ndraws, C.byref(sizes_c), rt._stream_ptr()), "vgx_flatten_count")
            if collect is not None:
                collect.update(dict(ctx.stage_times()))
            rt._check(L.vgx_flatten_emit(ctx.handle, pset.handle, dd.data_ptr(), ndraws, 1, C.byref(out), rt._stream_ptr()), "vgx_flatten_emit")
            if collect is not None:
                torch.cuda.synchronize()
                for k, v in ctx.stage_times():
                    collect[k] = collect.get(k, 0.0) + v

        def step1(collect=None):
            rt.flatten_async(ctx, pset, dd, ndraws, fb, apply_transform=True)
            if collect is not None:
                torch.cuda.synchronize()
                collect.update(dict(ctx.stage_times()))
        step = step2 if two_phase else step1
        if not two_phase:
            # the companion number: the two-phase entry on the same batch, a few steps, outside the timed region
            for _ in range(2):
                step2()
            torch.cuda.synchronize()
            tq = time.perf_counter()
            for _ in range(3):
                step2()
            torch.cuda.synchronize()
            res["two_phase_ms_per_step"] = (time.perf_counter() - tq) / 3 * 1e3
        units = npv
        unit_name = "polyline vertices"
        bufs = None
    else:
        tc0 = time.perf_counter()
        sizes = rt.tessellate_count(ctx, pset, dd, ndraws)
        res["setup_ms"]["first_count"] = round((time.perf_counter() - tc0) * 1e3, 3)
        bufs = rt.MeshBuffers(dev, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
        if placements > 1:
            # Where the caller's 8.75 GB of output buffers land in physical memory decides 5-25 % of the emit kernels' speed
            # (stable per allocation, independent of the virtual layout). A caller that keeps its output
            # buffers across frames can pick among a few allocations once; the bench does the same BEFORE the warm-up and
            # the timed region, and reports every candidate's time.
            cand = [bufs] + [rt.MeshBuffers(dev, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]) for _ in range(placements - 1)]
            probe_ms = []
            for b in cand:
                for _ in range(2):
                    rt.tessellate_async(ctx, pset, dd, ndraws, b)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    rt.tessellate_async(ctx, pset, dd, ndraws, b)
                torch.cuda.synchronize()
                probe_ms.append((time.perf_counter() - t0) / 3 * 1e3)
            pick = min(range(len(cand)), key=lambda i: probe_ms[i])
            bufs = cand[pick]
            res["output_placement"] = {"candidates": placements, "probe_ms_per_step": [round(x, 3) for x in probe_ms], "picked": pick}
            del cand, b
            torch.cuda.empty_cache()

        def step(collect=None):
            rt.tessellate_async(ctx, pset, dd, ndraws, bufs)
            if collect is not None:
                torch.cuda.synchronize()
                collect.update(dict(ctx.stage_times()))
        units = sizes["num_vertices"]
        unit_name = "output vertices"
    for _ in range(warmup):
        step()
    barrier()
    if bufs is not None:
        status = int(bufs.dev_status.item())
        assert status == 0, "device status %d" % status
    ctx.set_profiling(True)  # HIP events between kernels on the launch stream (cheap; part of the timed region)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if kind == "flatten":
        # two entry points per step with a host synchronisation inside the first: the per-kernel times come from a few extra steps
        samples = []
        for _ in range(min(5, max(1, steps))):
            c = {}
            step(c)
            samples.append(c)
        for c in samples:
            for k, v in c.items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v / len(samples)
    else:
        # per-kernel HIP-event durations of the TIMED steps themselves (the library keeps one event set per call, up to
        # VGX_PROF_RING = 32 calls back): back-to-back launches, no synchronisation in between
        stage_sum = dict(ctx.stage_times(ncalls=min(steps, 32)))
        res["stage_calls_averaged"] = min(steps, 32)
    ctx.set_profiling(False)
    # The same K steps once more, back to back with the timed region: `--warmup 5` of a 2 ms step is 10 ms of load, less than the
    # GPU's power management needs to reach its sustained clock from idle (the host-side CPU baseline runs first, the device idles
    # meanwhile). `value` stays what the contract defines (W warm-up steps, then K timed steps); this is reported beside it.
    barrier()
    t1 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    res_sustained = (time.perf_counter() - t1) / max(1, steps) * 1e3
    res["sustained_ms_per_step"] = res_sustained
    if kind == "flatten" and not two_phase:
        assert int(fb.dev_status.item()) == 0, "vgx_flatten device status %d" % int(fb.dev_status.item())
        gz = fb.dev_sizes.cpu().numpy()
        assert int(gz[0]) == npv and int(gz[1]) == nsp, ("vgx_flatten totals", int(gz[0]), npv, int(gz[1]), nsp)
    if kind != "flatten":
        # COLD step: vgx_tessellate_count + vgx_tessellate per step -- what a frame whose geometry changed costs.
        # The count call holds the flattener work of template batches (the first period is flattened and its tables are built there)
        # and, for every batch, the sizing passes and a host synchronisation; the steady-state step above re-uses its result.
        ncold = min(5, max(1, steps))
        rt.tessellate_count(ctx, pset, dd, ndraws)
        barrier()
        tcs = time.perf_counter()
        count_s = 0.0
        for _ in range(ncold):
            c0 = time.perf_counter()
            rt.tessellate_count(ctx, pset, dd, ndraws)
            count_s += time.perf_counter() - c0
            rt.tessellate_async(ctx, pset, dd, ndraws, bufs)
        barrier()
        res["cold_ms_per_step"] = (time.perf_counter() - tcs) / ncold * 1e3
        res["cold_count_ms"] = count_s / ncold * 1e3
        res["cold_steps"] = ncold
    # ONE-SHOT batch: everything a batch that is tessellated once pays, on a context whose scratch is warm (a
    # renderer keeps its context) -- new path set from the caller's arrays, the draw records up from pinned memory, the count
    # (output sizes: the caller has to allocate), one step, the wait. Outputs go to the buffers already allocated.
    try:
        nshot = 3
        shots = []
        for _ in range(nshot):
            torch.cuda.synchronize()
            o0 = time.perf_counter()
            pset1 = rt.PathSet(ctx, ps)
            o1 = time.perf_counter()
            dd1 = rt.upload_draws(pinned, local_rank)
            if kind == "flatten":
                rt._check(L.vgx_flatten_count(ctx.handle, pset1.handle, dd1.data_ptr(), ndraws, C.byref(sizes_c), rt._stream_ptr()), "vgx_flatten_count")
                o2 = time.perf_counter()
                rt.flatten_async(ctx, pset1, dd1, ndraws, fb, apply_transform=True)
            else:
                rt.tessellate_count(ctx, pset1, dd1, ndraws)
                o2 = time.perf_counter()
                rt.tessellate_async(ctx, pset1, dd1, ndraws, bufs)
            torch.cuda.synchronize()
            o3 = time.perf_counter()
            shots.append(((o3 - o0) * 1e3, (o1 - o0) * 1e3, (o2 - o1) * 1e3, (o3 - o2) * 1e3))
            pset1.close()
            del dd1
        best = min(shots)
        res["setup_ms"].update({"one_shot_ms": round(best[0], 3), "one_shot_split_ms": {"pathset_create": round(best[1], 3), "h2d_draws_and_count": round(best[2], 3), "step_and_wait": round(best[3], 3)},
                                "one_shot_runs_ms": [round(x[0], 3) for x in shots]})
        # leave the context as the timed steps left it (the count of the one-shot legs belongs to a path set that is gone)
        if kind == "flatten":
            rt._check(L.vgx_flatten_count(ctx.handle, pset.handle, dd.data_ptr(), ndraws, C.byref(sizes_c), rt._stream_ptr()), "vgx_flatten_count")
        else:
            rt.tessellate_count(ctx, pset, dd, ndraws)
            rt.tessellate_async(ctx, pset, dd, ndraws, bufs)
            torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001 -- a companion number must not take the config with it
        res["setup_ms"]["one_shot_error"] = repr(e)[:160]
    del pinned
    fill_verts = fill_idx = fill_meshes = 0
    if bufs is not None:
        assert int(bufs.dev_status.item()) == 0
        got = bufs.dev_sizes.cpu().numpy()
        assert int(got[3]) == sizes["num_vertices"] and int(got[4]) == sizes["num_indices"]
        sizes["num_fill_elements"] = int(got[8])
        sizes["num_elements"] = int(got[7])
        mt = bufs.meshes[:sizes["num_meshes"] * 32].view(torch.int32).view(-1, 8)
        is_fill = (mt[:, 7] >> 28) <= 1  # VGX_MESH_FILL / VGX_MESH_FILL_AA
        fill_verts = int(mt[is_fill, 4].to(torch.int64).sum().item())
        fill_idx = int(mt[is_fill, 5].to(torch.int64).sum().item())
        fill_meshes = int(is_fill.sum().item())
        del mt, is_fill
    else:
        sizes.setdefault("num_fill_elements", 0)
        sizes.setdefault("num_elements", 0)
    mode = ctx.failure_info()["segment_items"] if kind != "flatten" else 0  # which flatten kernel the library chose (0 command-parallel, 1 / 2 / 3 / 4 instanced, 5 template, 6 command-parallel with the static layout of lineTo-only path sets)
    res["flatten_mode"] = mode
    ab = algorithmic_bytes(cmd_bytes, ndraws, sizes, fill_verts, fill_idx, fill_meshes, instanced=mode not in (0, 5, 6), template=mode == 5)
    if kind == "flatten":
        ab["pipeline"] = ab["flatten_emit"]
        if "two_phase_ms_per_step" in res:
            res["two_phase_ms_per_step"] = round(res["two_phase_ms_per_step"], 3)
    res.update(dt=dt, sizes=sizes, stage=stage_sum, ab=ab, units=units, unit_name=unit_name, bufs=bufs, pset=pset, dd=dd, scratch=ctx.scratch_bytes())
    return res


def run_animated(rt, torch, ctx, local_rank, wl, instances, steps, warmup, barrier):
    """Tiger x`instances` with the drawing's geometry changing every step (an animated drawing): per step a NEW path set is created
    (host validation + derived tables + H2D), vgx_tessellate_count sizes the batch (template batches: flattens the first period and
    rebuilds the template -- the flattener is inside the timed region) and vgx_tessellate emits. Output capacity is fixed up front
    (the caller's job, as with the reference's grow-only buffers); the previous step's path set is destroyed at the start of the
    next step (hipFree synchronises: part of the price of not keeping it)."""
    import numpy as np
    from importlib import import_module
    psm = import_module("vg-renderer_amd.pathset")
    dev = torch.device("cuda", local_rank)
    ps0, ops = wl.tiger_paths()
    d = wl.tiger_draws(ops, instances, first_instance=0)
    ndraws = int(d.shape[0])
    dd = rt.upload_draws(d, local_rank)
    n = steps + warmup
    # every step's drawing: the control points breathe about the drawing's centre (scale 1 .. 1.03) -- subdivision counts change
    c = np.float32(450.0)
    variants = []
    for k in range(n):
        f = np.float32(1.0 + 0.03 * (0.5 - 0.5 * np.cos(2.0 * np.pi * k / max(2, n))))
        variants.append(psm.PathSetArrays(ps0.cmd_type, ps0.cmd_arg_off, ((ps0.args - c) * f + c).astype(np.float32), ps0.path_cmd_begin))
    pset = rt.PathSet(ctx, variants[-1])  # the largest-ish variant sizes the output buffers (+ 10 %)
    z = rt.tessellate_count(ctx, pset, dd, ndraws)
    pset.close()
    bufs = rt.MeshBuffers(dev, int(z["num_vertices"] * 1.1), int(z["num_indices"] * 1.1), int(z["num_meshes"] * 1.1) + 16)
    prev = [None]
    t_ps = t_cnt = 0.0
    verts = 0
    modes = set()

    def step(k, timed):
        nonlocal t_ps, t_cnt, verts
        a0 = time.perf_counter()
        if prev[0] is not None:
            prev[0].close()
        p = rt.PathSet(ctx, variants[k])
        a1 = time.perf_counter()
        sz = rt.tessellate_count(ctx, p, dd, ndraws)
        a2 = time.perf_counter()
        rt.tessellate_async(ctx, p, dd, ndraws, bufs)
        prev[0] = p
        if timed:
            t_ps += a1 - a0
            t_cnt += a2 - a1
            verts += int(sz["num_vertices"])
            modes.add(ctx.failure_info()["segment_items"])

    for k in range(warmup):
        step(k, False)
    barrier()
    t0 = time.perf_counter()
    for k in range(warmup, n):
        step(k, True)
    barrier()
    dt = time.perf_counter() - t0
    assert int(bufs.dev_status.item()) == 0, "device status %d" % int(bufs.dev_status.item())
    prev[0].close()
    ms = dt / steps * 1e3
    return {"config": WORKLOADS["tiger10k_animated"],
            "workload": "tiger-like drawing (seed 2024) x %d instances per GPU; every step the drawing's control points are scaled about its centre by a new factor in [1, 1.03]" % instances,
            "value": round(verts / dt / 1e6, 2), "unit": "M output vertices/s", "ms_per_step": round(ms, 3), "steps": steps,
            "verts_per_step_avg": verts // max(1, steps),
            "split_ms": {"pathset_destroy_create": round(t_ps / steps * 1e3, 3), "tessellate_count": round(t_cnt / steps * 1e3, 3),
                         "tessellate_and_wait": round(ms - (t_ps + t_cnt) / steps * 1e3, 3)},
            "flatten_modes_seen": sorted(modes)}


def frame_leg(rt, torch, ctx, local_rank):
    """ONE real frame: the tiger-like drawing recorded by the reference's own command-list writers
    (tests/golden/frame_tiger_x1.npz, made by tests/golden/make_frame_fixture.py) -> vgx_cmdlist_decode (host) -> path set + draw
    records uploaded -> vgx_tessellate_count -> vgx_tessellate with draw-command assembly armed, in microseconds, checked against
    what the reference's Context handed to bgfx for the same list; beside it the reference Context itself on one host core."""
    import numpy as np
    cm = importlib.import_module("vg-renderer_amd.cmdlist")
    fx = np.load(os.path.join(ROOT, "tests", "golden", "frame_tiger_x1.npz"))
    data = fx["bytes"].tobytes()
    kw = dict(mtx=[float(x) for x in fx["mtx"]], global_alpha=float(fx["global_alpha"]), tess_tol=float(fx["tess_tol"]), fringe=float(fx["fringe"]),
              canvas=(float(fx["canvas"][0]), float(fx["canvas"][1])), white_uv=[int(x) for x in fx["white_uv"]], font_image=int(fx["font_image"]))
    dev = torch.device("cuda", local_rank)
    extra = {}
    rc, ps, draws, n = cm.decode(rt, data, extra=extra, time_reps=50, **kw)
    assert rc == 0 and n["skipped"] == 0
    dec_us = extra["decode_seconds"] * 1e6
    t0 = time.perf_counter()
    pset = rt.PathSet(ctx, ps)
    dd = rt.upload_draws(draws, local_rank)
    torch.cuda.synchronize()
    up_us = (time.perf_counter() - t0) * 1e6
    nd = int(draws.shape[0])
    t0 = time.perf_counter()
    sizes = rt.tessellate_count(ctx, pset, dd, nd)
    cnt_us = (time.perf_counter() - t0) * 1e6
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    bufs = rt.MeshBuffers(dev, nv, ni, nm)
    cmds = torch.zeros((nm + 2) * 48, dtype=torch.uint8, device=dev)
    ncmd = torch.zeros(1, dtype=torch.int64, device=dev)
    uv = torch.zeros((max(nv, 1), 2), dtype=torch.int16, device=dev)
    ctx.set_assembly(cmds, 65536, ncmd, split_state=True, uv=uv, uv_value=(int(fx["white_uv"][0]), 0))
    out = {}
    try:
        for _ in range(5):
            rt.tessellate_async(ctx, pset, dd, nd, bufs)
        torch.cuda.synchronize()
        assert int(bufs.dev_status.item()) == 0
        # the frame equals the reference Context's (sizes and exact stream sums of the fixture; the byte-for-byte comparison is
        # tests/test_cmdlist_ref.py's)
        ok = (nv == int(fx["ref_num_vertices"]) and ni == int(fx["ref_num_indices"]) and int(ncmd.item()) == int(fx["ref_num_drawcmds"])
              and int(bufs.idx[:ni].to(torch.int64).bitwise_and(0xFFFF).sum().item()) == int(fx["ref_idx_sum"])
              and int(bufs.color[:nv].to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) == int(fx["ref_col_sum"])
              and float(bufs.pos[:nv].to(torch.float64).sum().item()) == float(fx["ref_pos_sum"]))
        R = 200
        t0 = time.perf_counter()
        for _ in range(R):
            rt.tessellate_async(ctx, pset, dd, nd, bufs)
        torch.cuda.synchronize()
        b2b_us = (time.perf_counter() - t0) / R * 1e6
        t0 = time.perf_counter()
        for _ in range(R):
            rt.tessellate_async(ctx, pset, dd, nd, bufs)
            torch.cuda.synchronize()
        sync_us = (time.perf_counter() - t0) / R * 1e6
        graph_us = None
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                rt.tessellate_async(ctx, pset, dd, nd, bufs)
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(R):
                g.replay()
            torch.cuda.synchronize()
            graph_us = (time.perf_counter() - t0) / R * 1e6
            del g
        except Exception as e:  # noqa: BLE001
            out["graph_error"] = repr(e)[:120]
        # a whole frame from the recorded bytes, nothing kept from the last one but the buffers: decode, new path set, uploads,
        # count, tessellate + assemble, wait
        Rw = 20
        t0 = time.perf_counter()
        for _ in range(Rw):
            rc2, ps2, dr2, _n2 = cm.decode(rt, data, **kw)
            p2 = rt.PathSet(ctx, ps2)
            d2 = rt.upload_draws(dr2, local_rank)
            rt.tessellate_count(ctx, p2, d2, nd)
            rt.tessellate_async(ctx, p2, d2, nd, bufs)
            torch.cuda.synchronize()
            p2.close()
        whole_us = (time.perf_counter() - t0) / Rw * 1e6
        # ... and as a renderer in steady state runs it: the context's scratch and the output capacities are what the last frames
        # needed (the reference's Path / Stroker / vertex buffers are grow-only too), so a frame is decode + new path set + uploads +
        # ONE asynchronous vgx_tessellate (capacities checked on the device: VGX_E_NOSPACE in dev_status would ask for a count) + wait
        t0 = time.perf_counter()
        for _ in range(Rw):
            rc2, ps2, dr2, _n2 = cm.decode(rt, data, **kw)
            p2 = rt.PathSet(ctx, ps2)
            d2 = rt.upload_draws(dr2, local_rank)
            rt.tessellate_async(ctx, p2, d2, nd, bufs)
            torch.cuda.synchronize()
            p2.close()
        whole1_us = (time.perf_counter() - t0) / Rw * 1e6
        assert int(bufs.dev_status.item()) == 0
        out["whole_frame_us_from_recorded_bytes_single_call"] = round(whole1_us, 1)
        # the same frame as a STATIC batch (vgx_set_static_batches: a retained command list re-submitted under a new camera): the count
        # flattens the list once, a frame is then the template emit kernel (+ the assembly kernels while armed)
        ctx.set_static_batches(True)
        try:
            t0 = time.perf_counter()
            rt.tessellate_count(ctx, pset, dd, nd)
            out["static_count_us"] = round((time.perf_counter() - t0) * 1e6, 1)
            bufs.pos.zero_(); bufs.idx.zero_(); bufs.color.zero_()
            for _ in range(5):
                rt.tessellate_async(ctx, pset, dd, nd, bufs)
            torch.cuda.synchronize()
            ok_static = (int(bufs.dev_status.item()) == 0 and int(ncmd.item()) == int(fx["ref_num_drawcmds"])
                         and int(bufs.idx[:ni].to(torch.int64).bitwise_and(0xFFFF).sum().item()) == int(fx["ref_idx_sum"])
                         and int(bufs.color[:nv].to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) == int(fx["ref_col_sum"])
                         and float(bufs.pos[:nv].to(torch.float64).sum().item()) == float(fx["ref_pos_sum"]))
            t0 = time.perf_counter()
            for _ in range(R):
                rt.tessellate_async(ctx, pset, dd, nd, bufs)
            torch.cuda.synchronize()
            out["static_tessellate_assembled_us_back_to_back"] = round((time.perf_counter() - t0) / R * 1e6, 1)
            t0 = time.perf_counter()
            for _ in range(R):
                rt.tessellate_async(ctx, pset, dd, nd, bufs)
                torch.cuda.synchronize()
            out["static_tessellate_assembled_us_with_sync"] = round((time.perf_counter() - t0) / R * 1e6, 1)
            ctx.set_assembly(None)
            for _ in range(5):
                rt.tessellate_async(ctx, pset, dd, nd, bufs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(R):
                rt.tessellate_async(ctx, pset, dd, nd, bufs)
            torch.cuda.synchronize()
            out["static_tessellate_meshes_only_us_back_to_back"] = round((time.perf_counter() - t0) / R * 1e6, 1)
            t0 = time.perf_counter()
            for _ in range(R):
                rt.tessellate_async(ctx, pset, dd, nd, bufs)
                torch.cuda.synchronize()
            out["static_tessellate_meshes_only_us_with_sync"] = round((time.perf_counter() - t0) / R * 1e6, 1)
            out["static_equals_reference_frame"] = bool(ok_static)
        except Exception as e:  # noqa: BLE001
            out["static_error"] = repr(e)[:160]
        finally:
            ctx.set_static_batches(False)
    finally:
        ctx.set_assembly(None)
    pset.close()
    out.update({"workload": "one frame: the tiger-like 240-path drawing (convexFillAA + strokes) recorded by the reference's vg::clXxx writers, %d bytes of commands -> %d draws, %d vertices, %d indices, %d draw command(s)" % (len(data), nd, nv, ni, int(ncmd.item())),
                "equals_reference_frame": bool(ok),
                "decode_us": round(dec_us, 1), "decode_MB_per_s": round(len(data) / extra["decode_seconds"] / 1e6, 1),
                "pathset_create_and_uploads_us": round(up_us, 1), "tessellate_count_us": round(cnt_us, 1),
                "tessellate_assembled_us_back_to_back": round(b2b_us, 1), "tessellate_assembled_us_with_sync": round(sync_us, 1),
                "tessellate_assembled_us_hip_graph_replay": None if graph_us is None else round(graph_us, 1),
                "whole_frame_us_from_recorded_bytes": round(whole_us, 1)})
    # the reference's own Context playing the same list on ONE host core (submitCommandList -> Path / Stroker -> vg::end)
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyvgref as RV
        if RV.available():
            with RV.RefContext(max_vb_vertices=65536) as rcx:
                cl = rcx.create_command_list(0)
                # replay the recorded bytes into a list of the reference: the fixture's script again
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from vgscript import Script, add_path
                wl = importlib.import_module("vg-renderer_amd.workloads")
                pst, ops = wl.tiger_paths()
                sc = Script()
                sc.push().translate(12.0, 7.0)
                for p, o in enumerate(ops):
                    add_path(sc, pst, p)
                    sc.fill(o["fill_color"], RV.fill_flags(True))
                    if o["stroke"]:
                        sc.stroke(o["stroke_color"], o["stroke_width"], RV.stroke_flags(0, 0, True))
                sc.pop()
                sc.play(rcx, cl)
                for _ in range(3):
                    rcx.begin(1280, 720, 1.0); rcx.op(RV.IMMEDIATE, RV.SubmitCommandList, (), (cl,)); rcx.lib.vgr_end(rcx.h); rcx.next_frame()
                Rc = 100
                t0 = time.perf_counter()
                for _ in range(Rc):
                    rcx.begin(1280, 720, 1.0); rcx.op(RV.IMMEDIATE, RV.SubmitCommandList, (), (cl,)); rcx.lib.vgr_end(rcx.h); rcx.next_frame()
                out["reference_context_us_per_frame_one_core"] = round((time.perf_counter() - t0) / Rc * 1e6, 1)
    except Exception as e:  # noqa: BLE001
        out["reference_context_error"] = repr(e)[:160]
    return out


def roofline(res, steps, traffic_for=None):
    """Roofline object of the dominant kernel (by HIP-event time) + the per-kernel table."""
    stage_sum, ab = res["stage"], res["ab"]
    cands = [k for k in stage_sum if k in ab and k != "pipeline"]
    dom = max(cands, key=lambda k: stage_sum[k])
    dom_ms = stage_sum[dom]
    achieved = ab[dom] / (dom_ms * 1e-3) / 1e9
    ms_per_step = res["dt"] / steps * 1e3
    traffic = None
    valu = None
    valu_active = None
    if traffic_for is not None:
        # HBM traffic of that kernel per launch: not measurable from inside the process; taken from the committed
        # rocprofv3 --pmc passes of this same command (profiles/traffic.json), only for the workload they were made on
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tj = json.load(f)
            if isinstance(traffic_for, str):
                traffic = tj["configs"][traffic_for]["kernels"][dom]["traffic_bytes"]
                valu = tj["configs"][traffic_for]["kernels"][dom].get("valu_insts")
                valu_active = tj["configs"][traffic_for]["kernels"][dom].get("valu_active_quads")
            elif tj.get("instances_per_gpu") == traffic_for and dom in tj["kernels"]:
                traffic = tj["kernels"][dom]["traffic_bytes"]
                valu = tj["kernels"][dom].get("valu_insts")
                valu_active = tj["kernels"][dom].get("valu_active_quads")
        except (OSError, ValueError, KeyError):
            pass
    # The second ruler ("VALU issue alongside"): vector instructions per launch (SQ_INSTS_VALU of the committed
    # --pmc pass of this same command) / the live kernel time, against what the chip can issue -- 256 CUs x 4 SIMDs, one wave64 VALU
    # instruction per SIMD every 2 cycles (MI355X_MICROARCH.md "Wave scheduling") at the 2.4 GHz maximum clock. For the kernels that HBM
    # says nothing about (k_flat1: 0.07 of the HBM peak) this is the ruler that applies.
    secondary = None
    if valu:
        ach = valu / (dom_ms * 1e-3) / 1e9
        secondary = {"bound": "valu_issue", "achieved": round(ach, 1), "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": round(ach / VALU_PEAK_GINST, 4),
                     "valu_insts_per_launch": valu}
        if valu_active:
            # SQ_ACTIVE_INST_VALU of the same pass (quad-cycles with a vector instruction executing, summed over the waves). On every kernel of
            # this library it comes to 1.00-1.05 quad-cycles per instruction: the counter's granularity, not a second measurement -- reported
            # as it is, no fraction derived from it (at 4 cycles per instruction every `frac` above would double)
            secondary["valu_active_quad_cycles_per_launch"] = valu_active
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "secondary": secondary,
            "traffic_ratio": None if traffic is None else round(traffic / ab[dom], 3),
            "kernel_ms": round(dom_ms, 3), "algorithmic_bytes": ab[dom],
            "by_kernel": {k: {"ms": round(stage_sum[k], 3), "achieved": round(ab[k] / (stage_sum[k] * 1e-3) / 1e9, 1),
                              "frac": round(ab[k] / (stage_sum[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                          for k in ("flatten_build", "flatten_count", "flatten_emit", "flatten_one_walk", "fill_emit", "stroke_emit", "tile_emit", "tmpl_verify", "tmpl_emit") if k in stage_sum and k in ab and stage_sum[k] > 0},
            "pipeline_achieved": round(ab["pipeline"] / (ms_per_step * 1e-3) / 1e9, 1)}


def _finite(x):
    """NaN / infinities -> None, recursively: the printed line must load with a strict JSON parser."""
    if isinstance(x, float):
        return x if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--instances", type=int, default=10000, help="Tiger instances PER GPU (BASELINE config: 10000)")
    ap.add_argument("--config", default="tiger10k", choices=sorted(k for k in WORKLOADS if k != "tiger10k_animated"), help="workload of the headline line (default: the BASELINE metric's)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs reported under 'configs' (1-GPU runs)")
    ap.add_argument("--gather", action="store_true", help="(default for --gpus > 1) also time the RCCL gather of the streams to rank 0")
    ap.add_argument("--no-gather", action="store_true", help="multi-GPU: skip the gather leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"), help="where the full record goes (every config, stage times, sweeps); stdout carries the short line only")
    ap.add_argument("--placements", type=int, default=1, help="output-buffer allocations to probe before the warm-up (default 1 = the first allocation is the one that is timed; "
                                                               "N > 1 reports every candidate and times the fastest -- a tuning aid, not the headline)")
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # VGX_BENCH_SHARE_GPU=1 (testing only): all ranks on cuda:0 with the gloo backend, to exercise the multi-rank code
    # path on a one-GPU box (RCCL refuses two ranks on one device)
    share_gpu = os.environ.get("VGX_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    rt = importlib.import_module("vg-renderer_amd.runtime")
    wl = importlib.import_module("vg-renderer_amd.workloads")

    cpu = None
    other_cpu = {}
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline()  # before any GPU work, in separate processes
        if not args.no_configs:
            for name, which in CONFIG_CPU.items():  # the reference beside every BASELINE config, on a bounded sample (4 s each)
                other_cpu[name] = cpu_baseline(budget_seconds=CONFIG_CPU_BUDGET.get(which, 2.5), which=which)
    genv = gpu_environment() if rank == 0 else None

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    K = args.instances
    ps, draws, workload_desc, kind = make_workload(wl, args.config, K, rank)
    if args.config in CONFIG_ENV:  # library options are read at vgx_create
        os.environ.update(CONFIG_ENV[args.config])
    ctx = rt.Context(local_rank)
    res = run_config(rt, torch, ctx, local_rank, args.config, ps, draws, kind, args.steps, args.warmup, barrier, placements=max(1, args.placements))
    del draws
    sizes, bufs, pset, dd, ndraws = res["sizes"], res["bufs"], res["pset"], res["dd"], res["ndraws"]
    dt = res["dt"]

    red_dev = torch.device("cpu") if share_gpu else dev
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    vtot = torch.tensor([float(res["units"])], dtype=torch.float64, device=red_dev)
    by_rank = None
    if world > 1:
        import torch.distributed as dist
        each = [torch.zeros(1, dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(each, tmax.clone())
        by_rank = [round(float(x.item()) / args.steps * 1e3, 3) for x in each]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vtot, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    res["dt"] = dt
    total_units = float(vtot.item())

    # ---- heterogeneous batch through vgx_partition (SURVEY 8e "for heterogeneous batches balance on the count-pass result") ----
    # Every rank holds the same batch (Tiger instances at 7 scales, sorted by scale: equal instance counts per rank would be
    # unbalanced), runs vgx_partition and tessellates its own range. Reported: per-rank times of the balanced and of the
    # equal-count split. One rank: the parts are timed one after the other on the one GPU (what each rank of an 8-GPU run would do).
    # Runs right after the headline's timing, while all ranks are still in step (the gather legs below can time out on a rank).
    hetero = None
    if args.config == "tiger10k" and not args.no_configs:
        try:
            hetero = hetero_leg(rt, torch, wl, dev, local_rank, rank, world, red_dev)
        except Exception as e:  # noqa: BLE001 -- a diagnostic leg must not take the headline line with it
            hetero = {"error": repr(e)}

    # ---- multi-GPU: the gather of the final streams to rank 0, timed by default, reported beside `value` ----
    # Preferred: libvgx's own RCCL gather behind the C-ABI (vgx_gather) on a dedicated communicator; the torch.distributed
    # version of the same layout (vg-renderer_amd/dist.py) when that cannot be set up (e.g. the shared-GPU test mode).
    # A watchdog thread bounds the C-ABI leg so that a stuck transfer can never swallow the bench line.
    gather_ms = None
    gather_via = None
    bail = False  # the C-ABI gather leg got stuck: print the line with what is measured and leave without touching the communicators again
    if (world > 1 or os.environ.get("VGX_BENCH_FORCE_GATHER") == "1") and not args.no_gather and bufs is not None:
        dm = importlib.import_module("vg-renderer_amd.dist")
        import threading
        if world == 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        box = {}

        def capi_leg():
            try:
                cg = dm.CapiGather(ctx, local_rank)
                allz = cg.sizes(sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"], ndraws)
                gb = None
                if rank == 0:
                    gb = rt.MeshBuffers(dev, sum(z.num_vertices for z in allz), sum(z.num_indices for z in allz), sum(z.num_meshes for z in allz))
                cg.gather(bufs, allz, 0, gb)  # warm-up: connection set-up is not part of a steady-state gather
                barrier()
                g0 = time.perf_counter()
                cg.gather(bufs, allz, 0, gb)
                barrier()
                box["ms"] = (time.perf_counter() - g0) * 1e3
                if rank == 0 and world == 1:  # one rank: the gathered streams are the local ones
                    nv = sizes["num_vertices"]
                    box["check"] = bool(torch.equal(gb.pos[:nv], bufs.pos[:nv]) and torch.equal(gb.meshes, bufs.meshes))
                # Overlapped form: the gather of step i runs on a second stream while step i + 1 tessellates
                # into the other output buffer. Nothing in vgx_gather touches tessellation scratch, so the two only meet in
                # the memory system and on the xGMI links. Steady-state rate over `steps` steps, gather included.
                try:
                    b2 = rt.MeshBuffers(dev, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
                    outs = [bufs, b2]
                    side = torch.cuda.Stream(device=dev)
                    main = torch.cuda.current_stream(dev)
                    done = [None, None]   # gather of the buffer finished: it may be overwritten
                    barrier()
                    o0 = time.perf_counter()
                    for it in range(args.steps):
                        ob = outs[it % 2]
                        if done[it % 2] is not None:
                            main.wait_event(done[it % 2])
                        rt.tessellate_async(ctx, pset, dd, ndraws, ob)
                        ready = torch.cuda.Event()
                        ready.record(main)
                        side.wait_event(ready)
                        with torch.cuda.stream(side):
                            cg.gather(ob, allz, 0, gb)
                            ev = torch.cuda.Event()
                            ev.record(side)
                        done[it % 2] = ev
                    side.synchronize()
                    barrier()
                    box["overlap_ms_per_step"] = (time.perf_counter() - o0) / args.steps * 1e3
                    del b2
                except Exception as e:  # noqa: BLE001 -- the un-overlapped number above stands
                    box["overlap_err"] = repr(e)
                # Tiled form ("emit chunk k + 1 while sending chunk k"): the frame is tessellated in T sub-batches of
                # whole instances, tile t into the local buffers behind tile t - 1; its gather (vgx_gather_at: place = the rank's
                # base + the tiles in front) runs on the second stream while tile t + 1 is tessellated. One frame's latency,
                # gather included, without a second set of output buffers. Tiles have the same shape (same drawing per instance),
                # so one count call sizes them all.
                # VGX_BENCH_GATHER_TILES: one tile count or a comma list swept in this run (default 1,2,4,8 with more than one rank);
                # the best one is reported as ms_per_step_with_tiled_gather, all of them in tiled_gather_sweep
                tile_counts = [int(x) for x in os.environ.get("VGX_BENCH_GATHER_TILES", "1,2,4,8" if world > 1 else "4").split(",") if x.strip()]
                sweep = {}
                for T in tile_counts:
                  try:
                    if args.config.startswith("tiger") and T >= 1 and K % T == 0 and ndraws % T == 0:
                          nd_t = ndraws // T
                          dd_tiles = [dd[t * nd_t * 64:(t + 1) * nd_t * 64] for t in range(T)]
                          st = rt.tessellate_count(ctx, pset, dd_tiles[0], nd_t)
                          tv, ti, tm = st["num_vertices"], st["num_indices"], st["num_meshes"]
                          if (tv * T, ti * T, tm * T) == (sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]):
                              views = [bufs.view(t * tv, tv, t * ti, ti, t * tm, tm) for t in range(T)]
                              RS = rt.capi.RankSizes
                              tile_all = (RS * world)(*[RS(tv, ti, tm, nd_t) for _ in range(world)])
                              places = []
                              for t in range(T):
                                  places.append((RS * world)(*[RS(r * tv * T + t * tv, r * ti * T + t * ti, r * tm * T + t * tm, r * ndraws + t * nd_t) for r in range(world)]))
                              gb2 = rt.MeshBuffers(dev, tv * T * world, ti * T * world, tm * T * world) if rank == 0 else None
                              side = torch.cuda.Stream(device=dev)
                              main = torch.cuda.current_stream(dev)

                              def tiled_frame():
                                  for t in range(T):
                                      rt.tessellate_async(ctx, pset, dd_tiles[t], nd_t, views[t])
                                      ready = torch.cuda.Event()
                                      ready.record(main)
                                      side.wait_event(ready)
                                      with torch.cuda.stream(side):
                                          cg.gather_at(views[t], tile_all, places[t], 0, gb2)
                                  fin = torch.cuda.Event()
                                  fin.record(side)
                                  main.wait_event(fin)
                              tiled_frame()  # warm-up
                              barrier()
                              o1 = time.perf_counter()
                              for it in range(args.steps):
                                  tiled_frame()
                              barrier()
                              t_ms = (time.perf_counter() - o1) / args.steps * 1e3
                              sweep[T] = round(t_ms, 3)
                              box["tiled_sweep"] = dict(sweep)
                              if box.get("tiled_ms_per_step") is None or t_ms < box["tiled_ms_per_step"]:
                                  box["tiled_ms_per_step"] = t_ms
                                  box["tiles"] = T
                              if rank == 0 and world == 1:
                                  nv = sizes["num_vertices"]
                                  box["tiled_check"] = bool(torch.equal(gb2.pos[:nv], bufs.pos[:nv]) and int(views[-1].dev_status.item()) == 0)
                              del gb2, views
                              rt.tessellate_count(ctx, pset, dd, ndraws)  # scratch back to the whole-frame shape
                  except Exception as e:  # noqa: BLE001
                    box["tiled_err"] = repr(e)
                    break
                del gb
                cg.close()
            except Exception as e:  # noqa: BLE001 -- any failure falls back to the torch.distributed gather
                box["err"] = repr(e)

        if os.environ.get("VGX_BENCH_GATHER", "capi") == "capi" and not share_gpu:
            th = threading.Thread(target=capi_leg, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get("VGX_BENCH_GATHER_TIMEOUT", "90")))
            if th.is_alive():
                box["err"] = "timeout"
                bail = True
        else:
            box["err"] = "disabled"
        ok = torch.tensor([1 if "ms" in box else 0], dtype=torch.int32, device=torch.device("cpu") if share_gpu else dev)
        if world > 1 and box.get("err") != "timeout":
            import torch.distributed as dist
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            gather_ms, gather_via = box["ms"], "vgx_gather (C-ABI, dedicated RCCL communicator)"
        elif box.get("err") != "timeout" and world > 1:
            barrier()
            g0 = time.perf_counter()
            gres = dm.gather_streams(bufs.pos, bufs.color, bufs.idx, bufs.meshes, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"], ndraws)
            barrier()
            gather_ms = (time.perf_counter() - g0) * 1e3
            gather_via = "torch.distributed isend/irecv (vg-renderer_amd/dist.py); C-ABI leg: %s" % box.get("err")
            del gres
        else:
            gather_via = "not measured: %s" % box.get("err")
        res["gather_check"] = box.get("check")
        res["overlap_ms_per_step"] = box.get("overlap_ms_per_step")
        res["overlap_err"] = box.get("overlap_err")
        for k in ("tiled_ms_per_step", "tiles", "tiled_check", "tiled_err", "tiled_sweep"):
            res[k] = box.get(k)

    # ---- next rows, measured beside the headline on rank 0 of a 1-GPU run: not part of `value` ----
    next_rows = None
    if rank == 0 and world == 1 and args.config == "tiger10k" and not args.no_configs:
        nv_all, ni_all = sizes["num_vertices"], sizes["num_indices"]
        # draw-command assembly armed: cost of the partition kernels inside one step
        cap = 2 * (nv_all // 65536) + 2
        cmds = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
        ncmd = torch.zeros(1, dtype=torch.int64, device=dev)
        ctx.set_assembly(cmds, 0, ncmd)
        ctx.set_profiling(True)
        rt.tessellate_async(ctx, pset, dd, ndraws, bufs)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            rt.tessellate_async(ctx, pset, dd, ndraws, bufs)
        torch.cuda.synchronize()
        asm_ms = (time.perf_counter() - t1) / 3 * 1e3
        asm_stage = dict(ctx.stage_times()).get("assemble")
        ctx.set_profiling(False)
        ctx.set_assembly(None)
        assert int(bufs.dev_status.item()) == 0
        # shape cache: ONE drawing tessellated, submitted K times (same transforms as the instances of the headline run)
        ps1, d1 = wl.tiger(1)
        dd1 = rt.upload_draws(d1, local_rank)
        s1 = rt.tessellate_count(ctx, pset, dd1, d1.shape[0])
        cb = rt.MeshBuffers(dev, s1["num_vertices"], s1["num_indices"], s1["num_meshes"])
        rt.tessellate_emit(ctx, pset, dd1, d1.shape[0], cb)
        cache = rt.MeshCache(ctx, cb, s1, dd1, d1.shape[0])
        inst = np.zeros(K, dtype=rt.capi.cache_instance_dtype)
        inst["num_meshes"] = cache.nm
        inst["mtx"][:, 0] = 1
        inst["mtx"][:, 3] = 1
        inst["mtx"][:, 4] = 37.0 * (np.arange(K) % 100)
        inst["mtx"][:, 5] = 41.0 * (np.arange(K) // 100)
        raw = torch.from_numpy(inst.view(np.uint8).reshape(-1).copy()).to(dev)
        for _ in range(2):
            rt.cache_submit(ctx, cache, raw, K, bufs)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(5):
            rt.cache_submit(ctx, cache, raw, K, bufs)
        torch.cuda.synchronize()
        cache_ms = (time.perf_counter() - t2) / 5 * 1e3
        assert int(bufs.dev_status.item()) == 0
        cache_bytes = 12 * nv_all + 2 * ni_all + 32 * sizes["num_meshes"]
        next_rows = {
            "draw_command_assembly": {"ms_per_step_armed": round(asm_ms, 3), "assemble_kernels_ms": None if asm_stage is None else round(asm_stage, 3),
                                      "draw_commands": int(ncmd.item()), "max_vb_vertices": 65536},
            "shape_cache_submit": {"value": round(nv_all / cache_ms / 1e3, 1), "unit": "M verts/s", "ms": round(cache_ms, 3),
                                   "achieved": round(cache_bytes / (cache_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "bound": "hbm",
                                   "frac": round(cache_bytes / (cache_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "workload": "one tiger-like drawing tessellated once, submitted %d times (vgx_cache_submit)" % K},
        }
        del cache, cb, raw
        try:
            next_rows["frame_tiger_x1"] = frame_leg(rt, torch, ctx, local_rank)
        except Exception as e:  # noqa: BLE001 -- a companion leg must not take the line with it
            next_rows["frame_tiger_x1"] = {"error": repr(e)[:200]}

    # ---- the other BASELINE configs north_star names ("N cubics, M-segment strokes"), 1-GPU runs, beside the headline ----
    other = None
    if rank == 0 and world == 1 and not args.no_configs:
        other = {}
        del bufs
        res["bufs"] = None
        pset.close()
        torch.cuda.empty_cache()
        for name in WORKLOADS:
            if name == args.config:
                continue
            if name == "tiger10k_animated":
                try:
                    other[name] = run_animated(rt, torch, ctx, local_rank, wl, K, min(args.steps, 8), 2, barrier)
                except Exception as e:  # noqa: BLE001 -- a companion leg must not take the line with it
                    other[name] = {"error": repr(e)}
                torch.cuda.empty_cache()
                continue
            ps2, d2, desc2, kind2 = make_workload(wl, name, K, 0)
            steps2 = min(args.steps, 5)
            ctx2 = ctx
            if name in CONFIG_ENV:  # library options are read at vgx_create: these configs get a context of their own
                saved = {k: os.environ.get(k) for k in CONFIG_ENV[name]}
                os.environ.update(CONFIG_ENV[name])
                ctx2 = rt.Context(local_rank)
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            r2 = run_config(rt, torch, ctx2, local_rank, name, ps2, d2, kind2, steps2, min(args.warmup, 2), barrier)
            ms2 = r2["dt"] / steps2 * 1e3
            other[name] = {"config": WORKLOADS[name], "workload": desc2,
                           "value": round(r2["units"] / (ms2 * 1e-3) / 1e6, 2), "unit": "M %s/s" % r2["unit_name"], "ms_per_step": round(ms2, 3), "ms_per_step_sustained": round(r2.get("sustained_ms_per_step", 0.0), 3), "steps": steps2,
                           "verts_per_gpu": r2["sizes"].get("num_vertices", 0), "indices_per_gpu": r2["sizes"].get("num_indices", 0),
                           "poly_verts_per_gpu": r2["sizes"]["num_poly_vertices"], "meshes_per_gpu": r2["sizes"].get("num_meshes", 0),
                           "flatten_kernel": {0: "k_flatten_build", 1: "k_flatten_inst", 2: "k_flatten_inst (grouped)", 3: "k_flatten_inst (grouped by path and tolerance class)", 4: "k_flatten_inst (instances sorted by tolerance class)", 5: "none per step (template mode: the class representatives flattened once by vgx_tessellate_count)", 6: "k_flatten_thin"}.get(r2.get("flatten_mode"), "k_flatten"),
                           "roofline": roofline(r2, steps2, traffic_for=name), "stage_ms": {k: round(v, 3) for k, v in r2["stage"].items()},
                           "setup_ms": r2.get("setup_ms"),
                           "cpu_baseline": other_cpu.get(name)}
            cb2 = other_cpu.get(name)
            if cb2 and other[name]["setup_ms"] is not None:
                # what the reference on all host cores needs for the same batch, once (from its measured rate on the bounded sample)
                other[name]["setup_ms"]["cpu_one_shot_ms"] = round(r2["units"] / (cb2["value"] * 1e6) * 1e3, 2)
            if r2.get("flatten_entry"):
                other[name]["entry"] = r2["flatten_entry"]
                if r2.get("two_phase_ms_per_step") is not None:
                    other[name]["two_phase_ms_per_step"] = r2["two_phase_ms_per_step"]
            if r2.get("cold_ms_per_step") is not None:
                other[name]["ms_per_step_cold"] = round(r2["cold_ms_per_step"], 3)
                other[name]["cold_count_ms"] = round(r2["cold_count_ms"], 3)
                other[name]["value_cold"] = round(r2["units"] / (r2["cold_ms_per_step"] * 1e-3) / 1e6, 2)
            r2["pset"].close()
            del r2, ps2, d2
            if name == "cubics1m":
                # "also run boxes 10 / 100 / 10 000 to sweep the output size" (coordinates uniform in
                # [0, box): ~5 / ~14 / ~145 segments per cubic against ~45 at box 1000). Parity at every box:
                # tests/test_gpu_fullsize_every_unit.py::test_cubics_box_sweep_matches_the_reference
                sweep = {}
                for box in (10.0, 100.0, 10000.0):
                    ps3, d3 = wl.random_cubics(1000000, seed=1234, box=box)
                    r3 = run_config(rt, torch, ctx2, local_rank, name, ps3, d3, "flatten", 3, 1, barrier)
                    ms3 = r3["dt"] / 3 * 1e3
                    sweep["%g" % box] = {"ms_per_step": round(ms3, 3), "poly_verts": r3["units"], "segments_per_cubic": round(r3["units"] / 1e6 - 1.0, 2),
                                         "value": round(r3["units"] / (ms3 * 1e-3) / 1e6, 2), "unit": "M polyline vertices/s",
                                         "two_phase_ms_per_step": r3.get("two_phase_ms_per_step"),
                                         "stage_ms": {k: round(v, 3) for k, v in r3["stage"].items()}}
                    r3["pset"].close()
                    del r3, ps3, d3
                    torch.cuda.empty_cache()
                other[name]["box_sweep"] = sweep
            if ctx2 is not ctx:
                ctx2.close()
            torch.cuda.empty_cache()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_units * args.steps / dt / 1e6
        metric = "M tessellated verts/sec (stroke+fill AA), Tiger×10k batch, 1/2/4/8 GPUs"
        if args.config != "tiger10k":
            metric = "M %s/sec, %s" % (res["unit_name"], WORKLOADS[args.config])
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "M verts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_desc, "name": args.config,
                       "instances_per_gpu": K if args.config == "tiger10k" else None, "draws_per_gpu": ndraws, "parallelism": "shard%d" % world,
                       "verts_per_gpu": sizes.get("num_vertices", 0), "indices_per_gpu": sizes.get("num_indices", 0), "meshes_per_gpu": sizes.get("num_meshes", 0),
                       "poly_verts_per_gpu": sizes["num_poly_vertices"], "serial_draws": sizes["num_serial_draws"],
                       "scratch_bytes_per_gpu": res["scratch"], "output_placement": res.get("output_placement"),
                       # which flatten kernel the 'flatten_build' stage is (the library decides per batch)
                       "flatten_kernel": ("none per step: template mode (the instances differ in transform / colours only; vgx_tessellate_count flattened the first period once, in local space)" if res.get("flatten_mode") == 5
                                          else "k_flatten_inst (one lane per instance: the draws repeat one sequence of paths)" if res.get("flatten_mode") == 1
                                          else "k_flatten_inst, grouped (draws sorted by path on the device)" if res.get("flatten_mode") == 2
                                          else "k_flatten_inst, grouped (draws sorted by path and tolerance class on the device)" if res.get("flatten_mode") == 3
                                          else "k_flatten_inst, periodic with the instances sorted by tolerance class on the device" if res.get("flatten_mode") == 4
                                          else "k_flatten_thin (lineTo-only path set: polyline layout decided when the set was created, gather - transform - scatter per batch)" if res.get("flatten_mode") == 6
                                          else "k_flatten_build (one lane per path command)")},
            "roofline": roofline(res, args.steps, traffic_for=K if args.config == "tiger10k" else args.config),
            "stage_ms": {k: round(v, 3) for k, v in res["stage"].items()},
            "cpu_baseline": cpu,
            "gpu_environment": genv,
            "next_rows": next_rows,
            "configs": other,
        }
        out["setup_ms"] = res.get("setup_ms")
        if cpu and out["setup_ms"] is not None:
            out["setup_ms"]["cpu_one_shot_ms"] = round(total_units / world / (cpu["value"] * 1e6) * 1e3, 2)
        if res.get("cold_ms_per_step") is not None and world == 1:
            # count + emit per step: the flattener and the sizing passes back inside what is reported
            out["ms_per_step_cold"] = round(res["cold_ms_per_step"], 3)
            out["cold_count_ms"] = round(res["cold_count_ms"], 3)
            out["value_cold"] = round(total_units / (res["cold_ms_per_step"] * 1e-3) / 1e6, 2)
        if hetero is not None:
            out["heterogeneous_partition"] = hetero
        if world > 1 and args.config == "tiger10k":
            # BASELINE.json configs[4] is "Tiger x80k sharded across 8 MI355X": 10k instances per GPU = this leg at N = 8
            out["config"]["baseline_config"] = ("configs[4]: Tiger x%dk in total, %d ranks" % (K * world // 1000, world)) if K * world == 80000 else ("configs[2] per GPU x %d ranks" % world)
        if res.get("sustained_ms_per_step") and world == 1:
            # the K steps that follow the timed K, back to back (clocks at their sustained level): not `value`, reported beside it
            out["ms_per_step_sustained"] = round(res["sustained_ms_per_step"], 3)
            out["value_sustained"] = round(total_units / (res["sustained_ms_per_step"] * 1e-3) / 1e6, 2)
        if by_rank is not None:
            out["ms_per_step_by_rank"] = by_rank
        if gather_ms is not None:
            # SURVEY 8e defines the scaling target INCLUDING the gather of the final streams to the root: `value` is the
            # tessellation rate of all ranks, `value_with_gather` the rate with one (un-overlapped) gather per step added
            out["gather_ms"] = round(gather_ms, 2)
            out["value_with_gather"] = round(total_units / ((dt / args.steps) + gather_ms * 1e-3) / 1e6, 2)
        if res.get("overlap_ms_per_step") is not None:
            # every rank runs its own pipeline; the slowest one is not reduced here (rank 0's clock, barriers on both sides)
            out["ms_per_step_with_overlapped_gather"] = round(res["overlap_ms_per_step"], 3)
            out["value_with_overlapped_gather"] = round(total_units / (res["overlap_ms_per_step"] * 1e-3) / 1e6, 2)
        elif res.get("overlap_err"):
            out["overlapped_gather_error"] = res["overlap_err"]
        if res.get("tiled_ms_per_step") is not None:
            # the frame tessellated in tiles, every tile gathered (vgx_gather_at) while the next one is tessellated
            out["ms_per_step_with_tiled_gather"] = round(res["tiled_ms_per_step"], 3)
            out["value_with_tiled_gather"] = round(total_units / (res["tiled_ms_per_step"] * 1e-3) / 1e6, 2)
            out["gather_tiles"] = res["tiles"]
            if res.get("tiled_sweep"):
                out["tiled_gather_sweep_ms"] = {str(k): v for k, v in sorted(res["tiled_sweep"].items())}
            if res.get("tiled_check") is not None:
                out["tiled_gather_check"] = res["tiled_check"]
        elif res.get("tiled_err"):
            out["tiled_gather_error"] = res["tiled_err"]
        if gather_via is not None:
            out["gather_via"] = gather_via
            if res.get("gather_check") is not None:
                out["gather_check"] = res["gather_check"]
        # ---- compact summary: every config's step time / rate / dominant kernel / roofline fraction in one small object. It is the
        # line's FIRST extra key and, once more, its LAST (the driver's record keeps the contract keys and the last 8 KB of stdout),
        # and the BASELINE configs' numbers also sit as flat scalars inside `config` (kept with the contract keys).
        def brief(name, ms, value, unit, rf, cold=None, extra=None):
            b = {"ms": ms, "value": value, "unit": unit}
            if rf:
                b.update({"kernel": rf.get("kernel"), "frac": rf.get("frac"), "traffic_ratio": rf.get("traffic_ratio")})
            if cold is not None:
                b["ms_cold"] = cold
            if extra:
                b.update(extra)
            return b
        summary = {args.config: brief(args.config, out["ms_per_step"], out["value"], out["unit"], out["roofline"], out.get("ms_per_step_cold"))}
        for name, o in (other or {}).items():
            if "error" in o:
                summary[name] = {"error": o["error"][:80]}
                continue
            summary[name] = brief(name, o["ms_per_step"], o["value"], o["unit"], o.get("roofline"), o.get("ms_per_step_cold"),
                                  {"split_ms": o["split_ms"]} if "split_ms" in o else None)
            if "box_sweep" in o:
                summary[name]["box_sweep_ms"] = {k: v["ms_per_step"] for k, v in o["box_sweep"].items()}
        fr1 = (next_rows or {}).get("frame_tiger_x1")
        if fr1 and "error" not in fr1:  # one real frame, microseconds (next_rows.frame_tiger_x1 has the workload and every step)
            summary["frame_tiger_x1"] = {k: fr1.get(k) for k in ("decode_us", "decode_MB_per_s", "tessellate_assembled_us_back_to_back", "tessellate_assembled_us_with_sync",
                                                                 "tessellate_assembled_us_hip_graph_replay", "static_tessellate_assembled_us_back_to_back", "static_tessellate_meshes_only_us_back_to_back", "static_equals_reference_frame",
                                                                 "whole_frame_us_from_recorded_bytes", "whole_frame_us_from_recorded_bytes_single_call", "reference_context_us_per_frame_one_core", "equals_reference_frame")}
            out["config"]["frame_tiger_x1_us_static_batch"] = fr1.get("static_tessellate_assembled_us_back_to_back")
            out["config"]["frame_tiger_x1_us_hip_graph_replay"] = fr1.get("tessellate_assembled_us_hip_graph_replay")
            out["config"]["frame_tiger_x1_us_reference_one_core"] = fr1.get("reference_context_us_per_frame_one_core")
            out["config"]["frame_tiger_x1_us_whole_from_bytes"] = fr1.get("whole_frame_us_from_recorded_bytes")
            out["config"]["frame_tiger_x1_us_whole_from_bytes_single_call"] = fr1.get("whole_frame_us_from_recorded_bytes_single_call")
        for name in ("cubics1m", "round10k", "round10k_static", "tiger10k_round", "tiger10k_culled", "tiger10k_animated", "tiger10k_command_parallel", "tiger10k_per_instance_flatten"):
            if name in summary and "ms" in summary[name]:
                out["config"]["%s_ms_per_step" % name] = summary[name]["ms"]
                if summary[name].get("frac") is not None:
                    out["config"]["%s_dominant_frac" % name] = summary[name]["frac"]
        if out.get("ms_per_step_cold") is not None:
            out["config"]["ms_per_step_cold"] = out["ms_per_step_cold"]
            out["config"]["value_cold"] = out["value_cold"]
        # ---- what is printed (round 6). The driver parses ONE JSON line from stdout; round 5's 34.5 KB line came back unparsed. The
        # line is now SHORT (< 8 KB, asserted by tests/test_gpu_bench_contract.py and here): the contract keys, `roofline`,
        # `cpu_baseline`, and the other configs' numbers as flat scalars inside `config`. Everything else (`configs`, `next_rows`,
        # `stage_ms`, sweeps, `summary`, the environment) goes to bench_details.json next to bench.py (--details PATH) and to stderr.
        full = dict(out)
        full["summary"] = summary
        rf = dict(out["roofline"])
        rf.pop("by_kernel", None)
        cfg = out["config"]
        for name, o in (other or {}).items():
            if isinstance(o, dict) and "error" not in o:
                cb = o.get("cpu_baseline")
                if cb:
                    cfg["%s_cpu_value" % name] = cb["value"]
                su = o.get("setup_ms") or {}
                if su.get("one_shot_ms") is not None:
                    cfg["%s_one_shot_ms" % name] = su["one_shot_ms"]
                if su.get("cpu_one_shot_ms") is not None:
                    cfg["%s_cpu_one_shot_ms" % name] = su["cpu_one_shot_ms"]
                r2f = o.get("roofline") or {}
                if name in ("cubics1m", "round10k") and r2f.get("traffic_ratio") is not None:
                    cfg["%s_traffic_ratio" % name] = r2f["traffic_ratio"]
                if name in ("cubics1m", "round10k") and isinstance(r2f.get("secondary"), dict):
                    cfg["%s_valu_issue_frac" % name] = r2f["secondary"].get("frac")
        if (res.get("setup_ms") or {}).get("one_shot_ms") is not None:
            cfg["one_shot_ms"] = res["setup_ms"]["one_shot_ms"]
            if res["setup_ms"].get("cpu_one_shot_ms") is not None:
                cfg["cpu_one_shot_ms"] = res["setup_ms"]["cpu_one_shot_ms"]
        for name, o in (other or {}).items():
            su = (o.get("setup_ms") or {}) if isinstance(o, dict) else {}
            if name in ("cubics1m", "round10k") and su.get("pathset_create") is not None:
                cfg["%s_pathset_create_ms" % name] = su["pathset_create"]
                cfg["%s_h2d_draws_GBps" % name] = su.get("h2d_draws_GBps")
        cpu_short = None
        if cpu is not None:
            cpu_short = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "sample", "value_sse_stroker", "single_core_value")}
        head_keys = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]
        short = {k: out[k] for k in head_keys}
        short["config"] = cfg
        short["roofline"] = rf
        short["cpu_baseline"] = cpu_short
        for k in ("ms_per_step_cold", "value_cold", "ms_per_step_sustained", "value_sustained", "ms_per_step_by_rank", "gather_ms", "value_with_gather",
                  "ms_per_step_with_overlapped_gather", "value_with_overlapped_gather", "ms_per_step_with_tiled_gather", "value_with_tiled_gather",
                  "gather_tiles", "gather_check", "tiled_gather_check"):
            if out.get(k) is not None:
                short[k] = out[k]
        if gather_via is not None:
            short["gather_via"] = gather_via[:120]
        if hetero is not None:
            short["heterogeneous_partition"] = {k: hetero.get(k) for k in ("balanced_max_over_min", "equal_count_max_over_min", "slowest_rank_gain", "error") if hetero.get(k) is not None}
        short["details"] = os.path.basename(args.details)
        short = _finite(short)
        line = json.dumps(short, allow_nan=False)
        if len(line) >= 8192:  # never again: drop the optional scalars rather than print a line the driver cannot read
            cfg = short["config"]
            for k in [k for k in cfg if k not in ("workload", "name", "instances_per_gpu", "draws_per_gpu", "parallelism", "verts_per_gpu", "indices_per_gpu",
                                                   "meshes_per_gpu", "cubics1m_ms_per_step", "round10k_ms_per_step", "cubics1m_dominant_frac", "round10k_dominant_frac")]:
                cfg.pop(k)
            line = json.dumps(short, allow_nan=False)
        assert len(line) < 8192, len(line)
        details = json.dumps(full)
        try:
            with open(args.details, "w") as f:
                f.write(details + "\n")
        except OSError as e:
            sys.stderr.write("bench_details: could not write %s: %r\n" % (args.details, e))
        try:  # RCCL writes a version banner to the C stdout at communicator creation: get it out BEFORE the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stderr.write("bench_details: " + details + "\n")
        sys.stderr.flush()
        print(line, flush=True)  # the LAST line of stdout
    if bail:
        os._exit(0)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
