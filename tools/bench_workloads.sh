#!/bin/bash
# the "other workloads" table of DESIGN.md section 5
cd "$(dirname "$0")/.."
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$*', '| value=%.3e'%d['value'], 'us/step=%.1f'%(1e3*d['ms_per_step']), 'alg GB/s=%.0f (%.1f%%)'%(r['achieved'], 100*r['frac']), 'valid=%.3f'%d['config']['valid_action_rate'], 'mean n=%.1f'%d['config']['mean_ring_len'])"; }
run --workload d1 --envs 4096
run --workload d1 --envs 32768
run --workload mixed --envs 32768
run --workload random --envs 32768
run --workload boundary0 --envs 65536
