#!/bin/bash
# usage: tools/resource_usage.sh [extra hipcc flags] -> one line per kernel: VGPRs | spilled SGPRs | spilled VGPRs | scratch B/lane | waves/SIMD
# (the shipped flags + -Rpass-analysis=kernel-resource-usage; device code only, nothing is linked)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fPIC --offload-arch=gfx950 \
  -mllvm -disable-machine-licm -mllvm -amdgpu-atomic-optimizer-strategy=None --cuda-device-only -c -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" reinforcementlearning4meshgeneration_amd/csrc/meshenv_hip.hip 2>&1 | python3 -c "
import re, sys, subprocess
rows, cur = {}, None
for ln in sys.stdin:
    m = re.search(r'Function Name: (\S+)', ln)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[a-z/A-Z]+\])?: (\d+) \[-Rpass', ln)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(['c++filt'] + list(rows), capture_output=True, text=True).stdout.split('\n')
print('# kernel | VGPRs | SGPRs spilled | VGPRs spilled | scratch B/lane | occupancy waves/SIMD')
for (k, r), n in sorted(zip(rows.items(), names), key=lambda t: t[1]):
    n = re.sub(r'\(.*\)$', '', n).replace('void ', '')
    print(f\"{n} | {r.get('VGPRs', '?')} | {r.get('SGPRs Spill', '?')} | {r.get('VGPRs Spill', '?')} | {r.get('ScratchSize', '?')} | {r.get('Occupancy', '?')}\")
"
