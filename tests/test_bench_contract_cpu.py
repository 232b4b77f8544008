"""CPU: the bench.py output contract, checked on the committed line of the round (profiles/r01_bench_line.json) and on
bench.py's own argument defaults -- the driver parses exactly these keys."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_keys():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_line.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("env-steps/sec") and base["metric"].startswith("env-steps/sec")
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.5
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    # value is consistent with the step time and the env count
    assert abs(d["value"] - d["config"]["n_envs_total"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the rocprof summary of the same command agrees with the live event timing within 5 %
    s = json.load(open(os.path.join(ROOT, "profiles", "r01_summary.json")))
    assert abs(s["kernel_trace"]["avg_ns"] / 1e3 - r["kernel_avg_us"]) / r["kernel_avg_us"] < 0.05
    assert s["kernel"].replace(" ", "") == r["kernel"].replace(" ", "")


def test_bench_defaults_finish_in_minutes_and_never_touch_the_reference():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1', src)
    assert re.search(r'"--steps", type=int, default=\d{2,3}\b', src) and re.search(r'"--warmup", type=int, default=\d{1,2}\b', src)
    assert "/root/reference" not in src
