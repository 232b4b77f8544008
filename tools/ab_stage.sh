#!/bin/bash
cd "$(dirname "$0")/.."
run() { python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-timing --envs $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f' % (1e3*d['ms_per_step']))"; }
for r in 1 2 3; do
 for n in 8192 12288 16384 32768; do
  a=$(run $n); b=$(MESHENV_LIGHT=1 run $n); c=$(MESHENV_LIGHT=0 run $n); d=$(MESHENV_LAZY=0 MESHENV_LIGHT=0 run $n)
  echo "n=$n default=$a light1=$b light0=$c lazy0=$d"
 done
done
