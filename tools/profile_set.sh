#!/bin/bash
# rocprofv3 passes (kernel trace + FETCH_SIZE + WRITE_SIZE + SQ counters, each in its own run) for every workload of the
# round besides the headline (tools/profile_round.sh does that one): gpurun_out/profset_<tag>/<key>/{trace,fetch,write,sq}
# and <key>/line.json (the driver's JSON line).  tools/summarize_set.py <tag> turns them into profiles/<tag>_<key>_*.
# usage: tools/profile_set.sh TAG [key ...]   (no keys: every workload)
tag=${1:-r04}
shift
only=" $* "
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/profset_$tag
mkdir -p $out
python3 $R/tools/source_state.py --require-fresh > $out/source_state.json || { echo "refusing to profile a stale library"; exit 1; }
cd /tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
run() {  # key, script and arguments...
  key=$1; shift
  if [ "$only" != "  " ] && [[ "$only" != *" $key "* ]]; then return; fi
  rm -rf $out/$key; mkdir -p $out/$key
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$key/trace -- python3 "$@" > $out/$key/line.json 2> $out/$key/trace.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$key/fetch -- python3 "$@" > /dev/null 2> $out/$key/fetch.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$key/write -- python3 "$@" > /dev/null 2> $out/$key/write.log
  rocprofv3 --pmc $SQ --output-format csv -d $out/$key/sq -- python3 "$@" > /dev/null 2> $out/$key/sq.log
  echo "$key done: $(tail -c 300 $out/$key/line.json | tr -d '\n' | cut -c1-200)"
}
B="--no-cpu-baseline --no-kernel-timing --steps 100 --warmup 20"
run d1       $R/bench.py --workload d1 $B
run mixed    $R/bench.py --workload mixed --envs 32768 $B
run random   $R/bench.py --workload random --envs 32768 $B
run big      $R/bench.py --envs 65536 $B
run actor1   $R/examples/policy_rollout.py --actor one-launch --steps 200 --warmup 40
run actor2   $R/examples/policy_rollout.py --actor fused --steps 200 --warmup 40
run move     $R/tools/bench_move.py 65536 60
run quality  $R/tools/bench_quality.py 65536 300
run domgen   $R/tools/bench_domgen.py 65536
run smooth   $R/tools/bench_smooth.py 65536 60
