"""CPU: the bench.py output contract, checked on the committed line of the round (profiles/r04_bench_line.json) and on
bench.py's own argument defaults -- the driver parses exactly these keys."""
import importlib.util
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_keys():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("env-steps/sec") and base["metric"].startswith("env-steps/sec")
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "N_envs=4096" in d["metric"] and "boundary()" in d["metric"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and "not HBM" in r["limited_by"]
    # no published number in BASELINE.md -> vs_baseline null; the ratio to the CPU baseline of the same run has its own key
    assert abs(d["vs_cpu_baseline"]["all_cores"] - d["value"] / d["cpu_baseline"]["all_cores"]["value"]) < 1e-9
    allc = d["cpu_baseline"]["all_cores"]
    assert allc["sched_getaffinity"] >= allc["cores"] >= 1
    # the all-cores leg runs at min(cgroup quota, affinity) threads and says so
    assert allc["cgroup_cpu_quota"] is None or allc["cores"] == min(allc["cgroup_cpu_quota"], allc["sched_getaffinity"])
    assert "one parallel region" in allc["openmp_schedule"]
    assert "log_capacity=0" in d["config"]["workload"] and d["config"]["log_capacity"] == 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.5
    # the guide's figure is the one printed as `traffic`; the calibrated and raw readings sit beside it
    assert abs(r["traffic_over_algorithmic"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 1e-9
    assert r["traffic_other_readings"]["raw_FETCH+WRITE"] < r["traffic_other_readings"]["calibrated"] < r["traffic"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert isinstance(c["cpu_model"], str) and c["cpu_model"] and c["fp64_transcendentals_per_env_step"]["atan2"] > 10
    i = r["instruction_side"]
    assert 0 < i["wait_frac"] < 1 and 0 < i["valu_active_frac"] < 1 and i["valu_insts_per_wave"] > 100
    # value is consistent with the step time and the env count
    assert abs(d["value"] - d["config"]["n_envs_total"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the rocprof summary of the same command agrees with the live event timing within 5 %
    s = json.load(open(os.path.join(ROOT, "profiles", "r04_summary.json")))
    assert abs(s["kernel_trace"]["avg_ns"] / 1e3 - r["kernel_avg_us"]) / r["kernel_avg_us"] < 0.05
    assert s["kernel"].replace(" ", "") == r["kernel"].replace(" ", "")


def test_bench_defaults_finish_in_minutes_and_never_touch_the_reference():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1', src)
    assert re.search(r'"--steps", type=int, default=\d{2,3}\b', src) and re.search(r'"--warmup", type=int, default=\d{1,2}\b', src)
    assert "/root/reference" not in src


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _closed_groups(n_launches, k):
    """Model of launch_step's bracketing (csrc/meshenv_hip.hip): launch i has pos = i % 2k; an event pair opens at
    pos == 0 and closes at pos == k - 1."""
    closed, open_ = 0, False
    for i in range(n_launches):
        pos = i % (2 * k)
        if pos == 0:
            open_ = True
        if pos == k - 1 and open_:
            closed, open_ = closed + 1, False
    return closed


@pytest.mark.parametrize("steps,warmup", [(20, 5), (1, 0), (200, 20), (2, 0), (30, 5), (49, 1), (1000, 0)])
def test_every_step_count_closes_a_timing_group(steps, warmup):
    """The driver's --steps 20 run of round 1 recorded no event group (group size 25 > 20) and lost its roofline."""
    bench = _load_bench()
    k = bench.timing_group(steps, 25)
    assert 1 <= k <= min(25, steps)
    assert _closed_groups(steps, k) >= 1


def test_roofline_is_unconditional_and_labels_follow_the_arguments():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'out["roofline"] = {' in src and "if kt is not None and len(kt):\n        alg" not in src
    assert '"timing": timing' in src and "wall (no HIP-event group recorded)" in src
    assert "cpu_model" in src and "fp64_transcendentals_per_env_step" in src and "instruction_side" in src
    assert 'coll = "RCCL" if args.backend == "nccl" else args.backend' in src
    assert "N_envs={n} per GPU" in src  # the metric string names the envs / workload actually run


def test_algorithmic_bytes_formula_matches_survey_8d():
    bench = _load_bench()
    # one failed step on a 30-ring: 28 * 30 + 158; one valid: + 28 * 30 + 48
    assert bench.algorithmic_bytes(1, 0, 30, 0) == 28 * 30 + 158
    assert bench.algorithmic_bytes(1, 1, 30, 30) == 56 * 30 + 206


def test_profile_summaries_refuse_a_stale_or_foreign_library(tmp_path, monkeypatch):
    """tools/source_state.py: a profile records the hash of the sources its library was built from; the summarisers refuse
    a recorded hash that is not the working tree's, and a library flagged older than its sources."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import source_state
    now = source_state.source_hash()
    assert len(now) == 64 and now == source_state.source_hash()
    with pytest.raises(SystemExit, match="re-profile"):
        source_state.check_recorded({"source_sha256": "0" * 64, "library_older_than_sources": False}, "prof_x")
    with pytest.raises(SystemExit, match="older than its sources"):
        source_state.check_recorded({"source_sha256": now, "library_older_than_sources": True}, "prof_x")
    # the committed headline profile names the sources it was taken on, and a commit
    s = json.load(open(os.path.join(ROOT, "profiles", "r04_summary.json")))
    assert len(s["source"]["source_sha256"]) == 64 and s["source"]["library_older_than_sources"] is False
    assert s["source"]["library_built_from"] == s["source"]["source_sha256"] and s["source"]["git_head"]
