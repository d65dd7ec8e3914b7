"""bench.py's output contract, on small batches: the one-rank line (metric, value, roofline with traffic_ratio, cpu_baseline
when asked, configs) and the multi-rank code path -- two ranks launched the way the driver launches them (torch.distributed.run,
one process per rank), both on cuda:0 over gloo (VGX_BENCH_SHARE_GPU=1: RCCL refuses two ranks on one device, the GPU test
boxes have one): weak scaling, max over ranks, per-rank times, the gather leg through vg-renderer_amd/dist.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strict(text):
    def bad(c):
        raise AssertionError("non-finite constant %s in the line" % c)
    return json.loads(text, parse_constant=bad)


def _line(out):
    """The LAST line of stdout is the record: short (< 8 KB: round 5's 34.5 KB line came back from the driver unparsed) and strict JSON."""
    last = out.rstrip("\n").splitlines()[-1]
    assert last.startswith("{") and len(last) < 8192, (len(last), last[:200])
    return _strict(last)


def test_one_rank_line(tmp_path):
    det = str(tmp_path / "details.json")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--instances", "300", "--no-cpu",
                                   "--placements", "1", "--details", det], text=True, timeout=600, cwd=ROOT)
    short = _line(out)
    # the short line: contract keys first, then roofline + cpu_baseline; the BASELINE configs readable from `config`
    assert list(short.keys())[:15] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                                       "config", "roofline", "cpu_baseline"]
    assert short["vs_baseline"] is None and short["data"] == "synthetic" and short["details"] == "details.json"
    assert set(short["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "kernel"} and "by_kernel" not in short["roofline"]
    for k in ("cubics1m_ms_per_step", "round10k_ms_per_step", "cubics1m_dominant_frac", "round10k_dominant_frac", "ms_per_step_cold", "value_cold",
              "cubics1m_one_shot_ms", "round10k_one_shot_ms"):
        assert short["config"][k] > 0, k
    assert "configs" not in short and "next_rows" not in short and "summary" not in short and "stage_ms" not in short
    with open(det) as f:
        d = json.load(f)
    for k in ("metric", "value", "unit", "ms_per_step", "n_gpus"):
        assert d[k] == short[k], k
    assert d["metric"].startswith("M tessellated verts/sec") and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "M verts/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["value"] > 0 and abs(d["value"] - d["config"]["verts_per_gpu"] / d["ms_per_step"] / 1e3) / d["value"] < 0.02
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert set(r["by_kernel"]) == {"tmpl_emit"} and r["kernel"] == "tmpl_emit"  # the headline batch is a template batch: one kernel per step
    assert d["config"]["flatten_kernel"].startswith("none per step: template mode")
    assert set(d["configs"]) == {"cubics1m", "round10k", "tiger10k_varied", "tigerspec10k", "tiger10k_per_instance_flatten", "tiger10k_command_parallel",
                                 "tiger10k_varied_per_instance_flatten", "tiger10k_open", "tiger10k_bevel", "tiger10k_round", "tiger10k_round_ordinary", "tiger10k_varied_round", "tiger10k_varied_round_ordinary", "cubics1m_stroked", "cubics1m_stroked_heap_route", "tiger10k_culled", "tiger10k_culled_ordinary", "round10k_static", "tiger10k_animated"}
    for name, c in d["configs"].items():
        assert "error" not in c, (name, c)
        assert c["value"] > 0, name
        if name != "tiger10k_animated":
            assert c["roofline"]["frac"] > 0, name
            assert set(c["setup_ms"]) >= {"pathset_create", "h2d_draws", "first_count"}, name
    # round 5: the flattener and the count are back inside what the line reports
    assert d["ms_per_step_cold"] > d["ms_per_step"] * 0.9 and d["value_cold"] > 0 and d["cold_count_ms"] > 0
    assert set(d["setup_ms"]) >= {"pathset_create", "h2d_draws", "first_count"}
    an = d["configs"]["tiger10k_animated"]
    assert set(an["split_ms"]) == {"pathset_destroy_create", "tessellate_count", "tessellate_and_wait"} and an["flatten_modes_seen"] == [5]  # the template is rebuilt every step
    assert d["configs"]["cubics1m"]["entry"].startswith("vgx_flatten (one walk") and d["configs"]["cubics1m"]["two_phase_ms_per_step"] > 0
    assert "flatten_one_walk" in d["configs"]["cubics1m"]["stage_ms"]
    assert set(d["summary"]) == set(d["configs"]) | {"tiger10k", "frame_tiger_x1"}
    f1 = d["next_rows"]["frame_tiger_x1"]
    assert "error" not in f1 and f1["equals_reference_frame"] is True and f1["decode_us"] > 0 and f1["tessellate_assembled_us_back_to_back"] > 0
    assert d["config"]["cubics1m_ms_per_step"] == d["configs"]["cubics1m"]["ms_per_step"] and d["config"]["round10k_ms_per_step"] == d["configs"]["round10k"]["ms_per_step"]
    assert d["configs"]["tiger10k_varied"]["flatten_kernel"].startswith("none per step")  # one template per scale class
    assert d["configs"]["tiger10k_varied_per_instance_flatten"]["flatten_kernel"] == "k_flatten_inst (instances sorted by tolerance class)"
    # the honesty configs really run the other pipelines
    assert d["configs"]["tiger10k_per_instance_flatten"]["flatten_kernel"] == "k_flatten_inst" and "tile_emit" in d["configs"]["tiger10k_per_instance_flatten"]["stage_ms"]
    assert d["configs"]["tiger10k_command_parallel"]["flatten_kernel"] == "k_flatten_build"
    assert d["configs"]["tigerspec10k"]["flatten_kernel"].startswith("none per step")
    assert d["configs"]["tiger10k_open"]["flatten_kernel"].startswith("none per step")  # open strokes: template mode's general kernel
    assert isinstance(d["gpu_environment"], dict)


def test_two_ranks_share_one_gpu(tmp_path):
    env = dict(os.environ, VGX_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                   "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                   "--instances", "200", "--no-cpu", "--placements", "1", "--details", str(tmp_path / "details2.json")], text=True, timeout=900, cwd=ROOT, env=env,
                                  stderr=subprocess.DEVNULL)
    d = _line(out)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert len(d["ms_per_step_by_rank"]) == 2 and abs(max(d["ms_per_step_by_rank"]) - d["ms_per_step"]) < 1e-2
    # whole-job value = both ranks' vertices over the slowest rank's time
    assert abs(d["value"] - 2 * d["config"]["verts_per_gpu"] / d["ms_per_step"] / 1e3) / d["value"] < 0.02
    assert d["gather_ms"] > 0 and d["value_with_gather"] < d["value"]
    assert "configs" not in d and d["cpu_baseline"] is None
