#!/bin/bash
# instruction-cache counters of the headline step kernel (own rocprofv3 pass, counters only)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_icache; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/a -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing "$@" > /dev/null 2> $out/a.log
python3 - "$(ls $out/a/*/*counter_collection.csv | head -1)" <<'PY'
import csv, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_step" not in k: continue
    per[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, d in per.items():
    m = len(n[k])
    print(k[:60], "launches", m, {c: round(v / m, 1) for c, v in sorted(d.items())})
PY
