#!/bin/bash
# rocprofv3 kernel trace of examples/policy_rollout.py (configs[2]: d1 envs + SAC-shaped actor in the loop)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_policy
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/examples/policy_rollout.py --steps 300 > $out/out.json 2> $out/trace.log < /dev/null
tail -n 1 $out/out.json
f=$(ls $out/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%")
PY
fi
