#!/bin/bash
cd "$(dirname "$0")/.."
for f in build_variants/lib_*.so; do echo $f; MESHENV_LIB=$PWD/$f python tools/rollout_sweep.py 2>&1 | grep -E "T= +64|T= 256"; MESHENV_LIB=$PWD/$f python bench.py --steps 100 --warmup 10 --envs 65536 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('  single-step 65536 envs: value=%.3e'%d['value'], 'us/step=%.2f'%(1e3*d['ms_per_step']))"; done
