#!/bin/bash
# instruction count of one valid rule-0 extraction (tools/pmc_valid_path.py); optional MESHENV_LIB=... for a variant
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_valid; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $out/a -- python3 $R/tools/pmc_valid_path.py 20 > $out/a.log 2>&1
tail -n 1 $out/a.log
python3 - "$(ls $out/a/*/*counter_collection.csv | head -1)" <<'PY'
import csv, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_step_group" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(per)[1:]            # the first launch picked the valid actions
A, B = ids[0::2], ids[1::2]
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"):
    a = sum(per[i][c] for i in A) / len(A); b = sum(per[i][c] for i in B) / len(B)
    print(f"{c[9:]:5s} per rejected wave {b / 4096:7.1f} | one valid rule-0 extraction adds {(a - b) / 256:8.1f} (check + update + reward helper)")
PY
