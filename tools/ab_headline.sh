#!/bin/bash
# A/B of build_variants/lib_*.so on the headline alone (4096 x boundary(), --steps 400), 4 interleaved rounds
cd "$(dirname "$0")/.."
for r in 1 2 3 4; do for f in build_variants/lib_*.so; do
  h=$(MESHENV_LIB=$PWD/$f python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('%.2f'%(1e3*json.loads(sys.stdin.readline())['ms_per_step']))")
  echo "$f headline_us=$h"
done; done
