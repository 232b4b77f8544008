"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the committed artefacts
profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.json and profiles/<tag>_bench_line.json.

usage: python tools/summarize_profile.py r01"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(os.path.join(src, pattern), recursive=True)
    if not files:
        raise SystemExit(f"nothing matches {pattern} under {src}")
    return max(files, key=os.path.getmtime)


def last_json_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


sys.path.insert(0, os.path.join(ROOT, "tools"))
import source_state
source = source_state.check_recorded(json.load(open(os.path.join(src, "source_state.json"))), "prof_" + tag)
stats_csv = newest("trace/**/*_kernel_stats.csv")
shutil.copy(stats_csv, os.path.join(dst, tag + "_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats_csv)))
# the one-step kernel the bench line names (the clock warm-up / pre-roll launches of the rollout kernel k_step<true, .> are
# in the trace too)
have_clean = os.path.exists(os.path.join(src, "bench.json")) and os.path.getsize(os.path.join(src, "bench.json")) > 0
want = last_json_line(os.path.join(src, "bench.json" if have_clean else "bench_under_rocprof.json"))["roofline"]["kernel"].replace(" ", "")
step = max((r for r in rows if want in r["Name"].replace(" ", "")), key=lambda r: float(r["TotalDurationNs"]))
kname = step["Name"].split("(")[0].replace("void ", "")


def pmc_means(sub):
    """Per-launch mean of every counter for the step kernel (a dispatch's rows are summed first: one row per
    counter instance)."""
    per = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(newest(sub + "/**/*_counter_collection.csv"))):
        if r["Kernel_Name"] == step["Name"]:
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {c: sum(d.values()) / len(d) for c, d in per.items()}


pmc = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    pmc.update(pmc_means(sub))
calib = None
cpath = os.path.join(dst, tag + "_fetch_calibration.json")
if os.path.exists(cpath):   # tools/calib_fetch.sh: what one counted FETCH_SIZE byte stands for in the step kernel's load mix
    calib = json.load(open(cpath))["factor_bytes_per_counted_byte"]
line_prof = last_json_line(os.path.join(src, "bench_under_rocprof.json"))
# (tools/profile_round.sh runs this script once on the GPU box BEFORE its clean bench run, so that the clean line's
#  roofline.traffic / instruction_side come from this very profile; bench.json does not exist yet then)
line = last_json_line(os.path.join(src, "bench.json")) if have_clean else line_prof
summary = {
    "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline",
    "workload": line["config"]["workload"],
    "kernel": kname,
    "kernel_trace": {"name": step["Name"][:120], "calls": int(step["Calls"]), "avg_ns": float(step["AverageNs"]),
                     "min_ns": float(step["MinNs"]), "max_ns": float(step["MaxNs"]),
                     "pct_of_gpu_time": float(step["Percentage"])},
    "pmc_per_launch_mean": dict(sorted(pmc.items())),
    "pmc_commands": [
        "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing",
        "rocprofv3 --pmc WRITE_SIZE -- (same)",
        "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -- (same)"],
    "hbm_traffic_bytes_per_launch": {
        "raw_(FETCH+WRITE)*1024": (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024,
        "gfx950_corrected_(2*FETCH+WRITE)*1024": (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024,
        "calibrated_(f*FETCH+WRITE)*1024": ((calib * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024) if calib else None,
        "fetch_calibration_factor": calib,
        "note": "MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of wide (16 B/lane) coalesced streams on gfx950; the ring "
                "loads here mix 16/8/4-byte-per-lane widths, so the corrected figure is an upper bound and the raw one "
                "a lower bound; the calibrated figure uses the factor measured for exactly this load mix on a footprint "
                "beyond every cache (tools/calib_fetch.sh -> profiles/<tag>_fetch_calibration.json)"},
    "n_envs_per_gpu": line["config"]["n_envs_per_gpu"],
    "workload_key": "boundary0",
    "source": source,
    "bench_line_under_rocprof": line_prof,
}
json.dump(summary, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
# the clean bench line is re-read AFTER the summary exists so that its roofline.traffic comes from this very profile
if have_clean:
    json.dump(line, open(os.path.join(dst, tag + "_bench_line.json"), "w"), indent=1)
ev = line["roofline"].get("kernel_avg_us")
print(f"{kname}: rocprof avg {float(step['AverageNs']) / 1e3:.2f} us over {step['Calls']} launches; bench events {ev} us; "
      f"value {line['value']:.4g} {line['unit']}; traffic <= {summary['hbm_traffic_bytes_per_launch']['gfx950_corrected_(2*FETCH+WRITE)*1024'] / 1e6:.2f} MB/launch")
