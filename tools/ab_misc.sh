#!/bin/bash
cd "$(dirname "$0")/.."
run() { MESHENV_LIB=$PWD/$1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-timing --workload $2 --envs $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f' % (1e3*d['ms_per_step']))"; }
for r in 1 2 3; do for f in build_variants/lib_*.so; do
  echo "$f d1_32768=$(run $f d1 32768) mixed_32768=$(run $f mixed 32768) b0_8192=$(run $f boundary0 8192) b0_1024=$(run $f boundary0 1024) b0_512=$(run $f boundary0 512)"
done; done
