#!/bin/bash
# FETCH_SIZE calibration for the step kernel's load mix -> gpurun_out/calib_fetch.json (copy to profiles/)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/calib_fetch; rm -rf $out; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_fetch $R/tools/calib_fetch.hip || exit 1
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc -- /tmp/calib_fetch 2097152 32 > $out/run.json 2> $out/run.log
python3 - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
info = json.loads(open(out + "/run.json").read().strip().splitlines()[-1])
per = {}
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_touch" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
vals = sorted(per.values())
fetch_kib = vals[len(vals) // 2]
res = dict(info, FETCH_SIZE_KiB_per_launch=fetch_kib, counted_bytes=fetch_kib * 1024,
           factor_bytes_per_counted_byte=info["bytes_requested_per_launch"] / (fetch_kib * 1024),
           note="load mix of load_env(): 16 + 4 + 8 + 4 B per lane over 32 lanes, 64 B record, 32 B counters, 12 B action, 72 B observation per env; "
                "footprint far beyond L2 + Infinity Cache, so every requested byte is fetched from HBM")
json.dump(res, open(out + "/../calib_fetch.json", "w"), indent=1)
print(json.dumps(res))
PY
