#!/bin/bash
# every build_variants/lib_*.so on: headline (4096), 65536 single-step, rollout T=256 -- interleaved, 3 rounds
cd "$(dirname "$0")/.."
for r in 1 2 3; do for f in build_variants/lib_*.so; do
  h=$(MESHENV_LIB=$PWD/$f python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('%.2f'%(1e3*json.loads(sys.stdin.readline())['ms_per_step']))")
  b=$(MESHENV_LIB=$PWD/$f python bench.py --steps 200 --warmup 20 --envs 65536 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('%.3e'%json.loads(sys.stdin.readline())['value'])")
  ro=$(MESHENV_LIB=$PWD/$f python tools/rollout_sweep.py 2>&1 | grep -E "T= 256" | sed -E 's/.*-> +([0-9.e+]+) env-steps.*/\1/')
  echo "$f headline_us=$h big=$b rollout256=$ro"
done; done
