#!/bin/bash
# SQ counters + kernel trace of the throughput-regime one-step kernel (65 536 x boundary()) -> gpurun_out/pmc_tp/summary.json
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_tp; rm -rf $out; mkdir -p $out
cd /tmp
A="--envs 65536 --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --envs 65536 --steps 100 --warmup 10 --no-cpu-baseline > $out/bench.json 2> $out/trace.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/sq -- python3 $R/bench.py $A > /dev/null 2> $out/sq.log
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/sq2 -- python3 $R/bench.py $A > /dev/null 2> $out/sq2.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/bench.py $A > /dev/null 2> $out/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/bench.py $A > /dev/null 2> $out/write.log
python3 - $out <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[1]
K = "k_step<false, true, false>"
agg = defaultdict(lambda: defaultdict(float))
for sub in ("sq", "sq2", "fetch", "write"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if K in r["Kernel_Name"]:
                agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
pmc = {c: sum(d.values()) / len(d) for c, d in sorted(agg.items())}
stats = [r for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True) for r in csv.DictReader(open(f)) if K in r["Name"]]
line = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
w = pmc["SQ_WAVES"]
res = dict(workload=line["config"]["workload"], kernel="meshenv::" + K, bench_value=line["value"], ms_per_step=line["ms_per_step"],
           kernel_trace=dict(calls=int(stats[0]["Calls"]), avg_ns=float(stats[0]["AverageNs"])) if stats else None,
           pmc_per_launch_mean=pmc,
           per_wave=dict(valu=pmc["SQ_INSTS_VALU"] / w, salu=pmc["SQ_INSTS_SALU"] / w, lds=pmc["SQ_INSTS_LDS"] / w),
           valu_issue_floor_us=pmc["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e3),
           valu_active_frac_of_wave_cycles=pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_WAVE_CYCLES"],
           wait_frac=pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"], issue_stall_frac=pmc["SQ_WAIT_INST_ANY"] / pmc["SQ_WAVE_CYCLES"],
           traffic_raw_bytes=(pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in ("bench_value", "ms_per_step", "kernel_trace", "per_wave", "valu_issue_floor_us", "valu_active_frac_of_wave_cycles", "wait_frac", "issue_stall_frac", "traffic_raw_bytes")}))
PY
