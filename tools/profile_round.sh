#!/bin/bash
# Collects the round's profile artefacts on the GPU box into gpurun_out/prof_<tag>/ (copy the summaries to profiles/).
tag=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_$tag
rm -rf $out
mkdir -p $out
python3 $R/tools/source_state.py --require-fresh > $out/source_state.json || { echo "refusing to profile a stale library"; exit 1; }
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $out/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $out/pmc_sq.log
cd $R
python3 tools/summarize_profile.py $tag > $out/summarize_stage1.log 2>&1   # profiles/<tag>_summary.json on the box: the clean run reads it
python3 bench.py --steps 200 --warmup 20 > $out/bench.json 2> $out/bench.log
tail -c 1500 $out/bench.json
