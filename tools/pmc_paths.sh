#!/bin/bash
# per-wave instruction counts of the step kernel for fixed action patterns
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_paths; rm -rf $out; mkdir -p $out
cd /tmp
for mode in memo rule0_out uniform; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $out/$mode -- python3 $R/tools/pmc_paths.py $mode 40 > $out/$mode.log 2>&1 < /dev/null
  f=$(ls $out/$mode/*/*counter_collection.csv 2>/dev/null | head -1)
  python3 - "$f" $mode <<'PY'
import csv, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_step" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(per)
def row(i):
    d = per[i]; w = d["SQ_WAVES"]
    return " ".join(f"{k[8:]}={d[k]/w:7.1f}" for k in sorted(d) if k != "SQ_WAVES")
print(sys.argv[2], "first launch :", row(ids[0]))
tail = ids[5:]
agg = defaultdict(float)
for i in tail:
    for k, v in per[i].items(): agg[k] += v
w = agg["SQ_WAVES"]
print(sys.argv[2], "later launches:", " ".join(f"{k[8:]}={agg[k]/w:7.1f}" for k in sorted(agg) if k != "SQ_WAVES"))
PY
done
