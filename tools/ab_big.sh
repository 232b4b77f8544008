#!/bin/bash
# A/B of build_variants/lib_*.so on the throughput regime: 65536 and 16384 boundary() envs, 32768 generated rings
cd "$(dirname "$0")/.."
for r in 1 2 3; do for f in build_variants/lib_*.so; do
  a=$(MESHENV_LIB=$PWD/$f python bench.py --steps 200 --warmup 20 --envs 65536 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('%.3e'%json.loads(sys.stdin.readline())['value'])")
  b=$(MESHENV_LIB=$PWD/$f python bench.py --steps 200 --warmup 20 --envs 16384 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('%.3e'%json.loads(sys.stdin.readline())['value'])")
  c=$(MESHENV_LIB=$PWD/$f python bench.py --steps 200 --warmup 20 --envs 32768 --workload random --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('%.3e'%json.loads(sys.stdin.readline())['value'])")
  echo "$f 65536=$a 16384=$b random32768=$c"
done; done
