#!/bin/bash
# SQ counters of the single-step kernel for the current library, MESHENV_SPEC=1 vs 0 -> gpurun_out/pmc_ab/
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_ab; rm -rf $out; mkdir -p $out
cd /tmp
for spec in 1 0; do  # MESHENV_SPEC=1: k_step_spec, 0: k_step_group
  export MESHENV_SPEC=$spec
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/sq_$spec -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $out/sq_$spec.log
  rocprofv3 --pmc SQ_IFETCH SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/sq2_$spec -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $out/sq2_$spec.log
done
python3 - $out <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
for spec in (1, 0):
    agg = defaultdict(lambda: defaultdict(float))
    for sub in ("sq", "sq2"):
        for f in glob.glob(f"{out}/{sub}_{spec}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_step_" in r["Kernel_Name"] and "ILb1E" not in r["Kernel_Name"].split("k_step")[1][:6]:
                    agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    print("spec" if spec else "group", {c: round(sum(d.values()) / len(d)) for c, d in sorted(agg.items())})
PY
