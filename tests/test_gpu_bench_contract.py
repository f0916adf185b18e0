"""bench.py's contract with the driver: one JSON line on stdout with the required keys, `roofline` and (unless switched
off) `cpu_baseline`; and the N > 1 code path (sc_allgather_paths + its consistency checks) driven at world size 1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _run(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--only-main-map",
                        "--replan-frames", "0", "--repeats", "3"] + extra, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines            # exactly one line on stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run(["--cpu-seconds", "1"])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["unit"] == "plans/s" and d["value"] > 0
    assert abs(d["value"] - d["config"]["queries_total"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["frac"] > 0.3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["gpu_matches_cpu_on_sample"] is True
    # the steps of a call are distinct batches (own grid, own queries): a step other than step 0 is checked against the oracle too
    assert c["gpu_matches_cpu_on_step"]["step_of_the_call"] == 3 and c["gpu_matches_cpu_on_step"]["matches"] is True
    assert d["repeats"] >= 3 and d["value_min"] <= d["value"] <= d["value_max"] and len(d["ms_per_region"]) == d["repeats"]
    assert d["config"]["queries_per_launch"] == 4 * d["config"]["queries_per_gpu"] and d["value_depth1"] > 0
    assert d["per_query_depth1"]["kilocycles_max"] >= d["per_query_depth1"]["kilocycles_p99"] >= d["per_query_depth1"]["kilocycles_p50"] > 0


def test_bench_gather_path_world1():
    d = _run(["--no-cpu-baseline"], env={"SC_BENCH_FORCE_DIST": "1"})
    g = d["gather"]
    assert g["consistent_on_all_ranks"] is True and g["truncated"] == 0
    assert 0 < g["bytes_received_per_rank_per_step"] < g["fixed_stride_bytes_per_rank_per_step"]
