"""gpurun_out/profset_<tag>/<key>/ (tools/profile_set.sh) -> profiles/<tag>_<key>_kernel_stats.csv + profiles/<tag>_<key>_summary.json:
per kernel of interest the rocprofv3 average duration, the algorithmic bytes per launch its driver states, the HBM counter
traffic (FETCH_SIZE / WRITE_SIZE per launch, raw and with the gfx950 correction of MI355X_MICROARCH.md) and the SQ counters
per wave.   usage: python tools/summarize_set.py r03"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "profset_" + tag)
dst = os.path.join(ROOT, "profiles")


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except json.JSONDecodeError:
                continue   # a Python dict printed by the driver, not its JSON line
    return {}


def counters(d, sub):
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> sum
    files = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)
    # gpurun merges into gpurun_out/ without deleting: an earlier run's files may sit beside this one's -- newest only
    for f in sorted(files, key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {c: sum(v.values()) / len(v) for c, v in cs.items()} for k, cs in per.items()}


sys.path.insert(0, os.path.join(ROOT, "tools"))
import source_state
source = source_state.check_recorded(json.load(open(os.path.join(src, "source_state.json"))), "profset_" + tag)

for key in sorted(os.listdir(src)):
    d = os.path.join(src, key)
    if not os.path.isdir(d):
        continue
    stats = sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]
    if not stats:
        print(key, "no kernel trace"); continue
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_{key}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats[0])))
    line = last_json(os.path.join(d, "line.json"))
    want = list(line.get("profile_kernels", []))
    if "roofline" in line:   # a bench.py line: the step kernel
        want.append({"match": line["roofline"]["kernel"].replace("meshenv::", "").split("<")[0] + "<",
                     "exact": line["roofline"]["kernel"], "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
                     "note": "SURVEY 8d: 28 sum_n + 158 steps + 28 sum_n_valid + 48 valid from the work counters of the run"})
    pmc = {}
    for sub in ("fetch", "write", "sq"):
        for k, cs in counters(d, sub).items():
            pmc.setdefault(k, {}).update(cs)
    out = {"tag": tag, "key": key, "source": source, "driver_line": {k: v for k, v in line.items() if k != "profile_kernels"}, "kernels": []}
    for w in want:
        cand = [r for r in rows if w["match"] in r["Name"] and (("exact" not in w) or w["exact"].replace(" ", "") in r["Name"].replace(" ", ""))]
        if not cand:
            continue
        r = max(cand, key=lambda r: float(r["TotalDurationNs"]))
        p = pmc.get(r["Name"], {})
        avg_ns = float(r["AverageNs"])
        alg = float(w["algorithmic_bytes_per_launch"])
        k = {"kernel": r["Name"][:100], "calls": int(r["Calls"]), "avg_us": avg_ns / 1e3, "min_us": float(r["MinNs"]) / 1e3,
             "pct_of_gpu_time": float(r["Percentage"]), "algorithmic_bytes_per_launch": alg, "algorithmic_note": w.get("note", ""),
             "achieved_GBps": alg / avg_ns, "frac_of_8TBps": alg / avg_ns / 8000.0}
        if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
            raw = (p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
            cor = (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
            k["hbm_traffic_bytes_per_launch"] = {"raw_(FETCH+WRITE)*1024": raw, "gfx950_corrected_(2*FETCH+WRITE)*1024": cor,
                                                 "corrected_over_algorithmic": cor / alg if alg else None}
        if "SQ_WAVES" in p and p["SQ_WAVES"] > 0:
            wv, cyc = p["SQ_WAVES"], p.get("SQ_WAVE_CYCLES", 0.0)
            k["sq_per_launch"] = {c: p[c] for c in sorted(p) if c.startswith("SQ_")}
            k["per_wave"] = {"valu": p["SQ_INSTS_VALU"] / wv, "salu": p["SQ_INSTS_SALU"] / wv, "lds": p["SQ_INSTS_LDS"] / wv}
            if cyc:
                k["wait_frac"] = p["SQ_WAIT_ANY"] / cyc; k["valu_active_frac"] = p["SQ_ACTIVE_INST_VALU"] / cyc
                k["issue_stall_frac"] = p["SQ_WAIT_INST_ANY"] / cyc
            k["valu_issue_floor_us"] = p["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e3)
        out["kernels"].append(k)
    json.dump(out, open(os.path.join(dst, f"{tag}_{key}_summary.json"), "w"), indent=1)
    for k in out["kernels"]:
        t = k.get("hbm_traffic_bytes_per_launch", {})
        print(f"{key:8s} {k['kernel'][:48]:48s} avg {k['avg_us']:9.2f} us  alg {k['algorithmic_bytes_per_launch'] / 1e6:8.2f} MB  "
              f"{k['achieved_GBps']:7.0f} GB/s ({100 * k['frac_of_8TBps']:.1f} %)  traffic x{(t.get('corrected_over_algorithmic') or 0):.2f}  "
              f"VALU/wave {k.get('per_wave', {}).get('valu', 0):.0f}")
