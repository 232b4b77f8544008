#!/bin/bash
# A/B of build_variants/lib_*.so on the long-ring workloads
cd "$(dirname "$0")/.."
for r in 1 2; do for f in build_variants/lib_*.so; do for w in "d1 4096" "d1 32768" "mixed 32768" "mixed 4096"; do set -- $w
  MESHENV_LIB=$PWD/$f python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-timing --workload $1 --envs $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$f $1 $2', 'value=%.3e'%d['value'], 'us/step=%.2f'%(1e3*d['ms_per_step']))"; done; done; done
