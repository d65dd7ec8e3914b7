"""Turn rocprofv3 (ROCm 7.2, rocpd SQLite output) result databases into the small text summaries that are
committed under profiles/. Usage:
  python profiles/summarize.py trace gpurun_out/prof_rNN_trace/trace_results.db > profiles/rNN_kernel_stats.txt
  python profiles/summarize.py pmc   gpurun_out/prof_rNN_fetch/fetch_results.db FETCH_SIZE > profiles/rNN_pmc_fetch.txt
"""
import sqlite3
import sys


def trace(db):
    c = sqlite3.connect(db).cursor()
    print("# rocprofv3 --kernel-trace --stats summary (durations in us)")
    print("%-100s %6s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 25"):
        print("%-100s %6d %14.1f %12.1f %6.2f%%" % (name[:100], calls, total, avg, pct))
    print()
    print("# launch geometry / registers of the vgx kernels")
    seen = set()
    for r in c.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels"):
        if r[0] in seen or ("k_" not in r[0]):
            continue
        seen.add(r[0])
        print("%-90s grid=%d wg=%d lds=%d scratch=%d vgpr=%d agpr=%d sgpr=%d" % ((r[0][:90],) + tuple(r[1:])))


def pmc(db, counter):
    c = sqlite3.connect(db).cursor()
    print("# rocprofv3 --pmc %s, per-kernel average over dispatches (value is KB as rocprofv3 reports it)" % counter)
    print("# gfx950 note (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by 2x;")
    print("# 'x2' column applies that correction for FETCH_SIZE; WRITE_SIZE is uncalibrated and shown raw.")
    rows = c.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name=? group by kernel_name order by avg(value)*count(*) desc limit 20", (counter,)).fetchall()
    print("%-90s %6s %14s %14s %12s" % ("kernel", "calls", "avg_KB", "avg_MB(x2)" if counter == "FETCH_SIZE" else "avg_MB", "avg_us"))
    for name, n, kb, dur in rows:
        mb = kb / 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
        print("%-90s %6d %14.1f %14.1f %12.1f" % (name[:90], n, kb, mb, dur / 1000.0))


def traffic(fetch_db, write_db, label, valu_db=None):
    """JSON for bench.py's roofline.traffic: HBM bytes per launch of the three big kernels = FETCH_SIZE x 2 (gfx950
    correction for wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE (raw), from two separate --pmc passes."""
    import json
    names = {"k_tmpl_emit": "tmpl_emit", "k_tmpl_emit_general": "tmpl_emit", "k_tmpl_emit_open": "tmpl_emit", "k_tmpl_emit_round": "tmpl_emit", "k_tmpl_emit_round_aa": "tmpl_emit", "k_tmpl_emit_round_aa_open": "tmpl_emit", "k_tmpl_emit_round_closed": "tmpl_emit", "k_tmpl_emit_bevel": "tmpl_emit", "k_tmpl_round_sizes": "tmpl_round_sizes", "k_tmpl_round_sizes_inst": "tmpl_round_sizes", "k_tmpl_round_sizes_block": "tmpl_round_sizes", "k_flatten_build": "flatten_build", "k_flatten_inst": "flatten_build", "k_fill": "fill_emit", "k_stroke": "stroke_emit", "k_stroke_long": "stroke_emit", "k_stroke_simple": "stroke_emit", "k_flatten_gather": "flatten_gather", "k_flatten_gather_ordered": "flatten_gather", "k_mesh_prepare": "mesh_prepare",
             "k_flatten<false": "flatten_count", "k_flatten<true": "flatten_emit",  # vgx_flatten_count / _emit (two walks): count pass, emit pass
             "k_flat1": "flatten_one_walk", "k_f1_seg_table": "flatten_one_walk",
             "k_emit_tiles": "tile_emit", "k_tile_table": "tile_emit"}  # round 6: the tile kernel of ordinary batches (+ its tile table)   # vgx_flatten (cubics1m, round 5): the one-walk kernel (+ its segment table; the REDO instance exits at once)
    out = {"source": label, "instances_per_gpu": 10000, "kernels": {}}
    for db, ctr, mul in ((fetch_db, "FETCH_SIZE", 2.0), (write_db, "WRITE_SIZE", 1.0)):
        c = sqlite3.connect(db).cursor()
        for name, kb in c.execute("select kernel_name, avg(value) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
            for k, stage in names.items():
                if k + "(" in name or k + "<" in name or ("<" in k and k in name):
                    d = out["kernels"].setdefault(stage, {"fetch_bytes": 0, "write_bytes": 0})
                    d["fetch_bytes" if ctr == "FETCH_SIZE" else "write_bytes"] += int(kb * 1024 * mul)  # flatten_build = k_flatten_inst + the k_flatten_build launch that exits at once (or the other way round); stroke_emit = k_stroke_simple + k_stroke likewise
    if valu_db:  # round 6: the second ruler (SURVEY 8d "VALU issue alongside"): vector instructions per launch, a --pmc SQ_INSTS_VALU pass of its own
        c = sqlite3.connect(valu_db).cursor()
        # SQ_INSTS_VALU: wave-instructions; SQ_ACTIVE_INST_VALU: quad-cycles (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles) during which
        # a wave has a vector instruction executing, summed over the waves -- collected in the same pass since round 6's last profile
        for ctr, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_ACTIVE_INST_VALU", "valu_active_quads")):
            for name, v in c.execute("select kernel_name, avg(value) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
                for k, stage in names.items():
                    if k + "(" in name or k + "<" in name or ("<" in k and k in name):
                        d = out["kernels"].setdefault(stage, {"fetch_bytes": 0, "write_bytes": 0})
                        d[key] = d.get(key, 0) + int(v)
    for d in out["kernels"].values():
        d["traffic_bytes"] = d["fetch_bytes"] + d["write_bytes"]
    print(json.dumps(out, indent=1))


def merge(label, parts):
    """profiles/traffic.json: the headline set at the top level (bench.py's headline line), every config's under "configs"."""
    import json
    out = {"source": label, "instances_per_gpu": 10000, "kernels": {}, "configs": {}}
    for part in parts:
        cfg, path = part.split("=", 1)
        with open(path) as f:
            t = json.load(f)
        out["configs"][cfg] = {"kernels": t["kernels"]}
        if cfg == "tiger10k":
            out["kernels"] = t["kernels"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "merge":
        merge(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "trace":
        trace(sys.argv[2])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
    else:
        pmc(sys.argv[2], sys.argv[3])
