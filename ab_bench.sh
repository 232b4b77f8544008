#!/bin/bash
# runs bench.py (no cpu baseline) for every build_variants/lib_*.so, 3 interleaved rounds
cd "$(dirname "$0")"
for r in 1 2 3; do
for f in build_variants/lib_*.so; do
  MESHENV_LIB=$PWD/$f python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$f', 'value=%.3e'%d['value'], 'ms/step=%.4f'%d['ms_per_step'], 'kern_us=%.2f'%d['roofline']['kernel_avg_us'], 'min=%.2f'%d['roofline']['kernel_min_us'])"
done; done
