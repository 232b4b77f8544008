#!/bin/bash
cd "$(dirname "$0")/.."
for r in 1 2 3; do for f in build_variants/lib_*.so; do
  MESHENV_LIB=$PWD/$f python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-timing --workload d1 --envs 4096 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$f d1 4096', 'value=%.3e'%d['value'], 'us/step=%.2f'%(1e3*d['ms_per_step']))"; done; done
