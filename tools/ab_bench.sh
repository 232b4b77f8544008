#!/bin/bash
# runs bench.py (no cpu baseline, no event timing) for every build_variants/lib_*.so, 3 interleaved rounds
cd "$(dirname "$0")/.."
for r in 1 2 3; do
for f in build_variants/lib_*.so; do
  MESHENV_LIB=$PWD/$f python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$f', 'value=%.3e'%d['value'], 'us/step=%.2f'%(1e3*d['ms_per_step']))"
done; done
