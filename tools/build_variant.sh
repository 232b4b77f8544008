#!/bin/bash
# usage: tools_ab.sh name "extra hipcc flags"   -> build_variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fPIC -shared --offload-arch=gfx950 -mllvm -disable-machine-licm -mllvm -amdgpu-atomic-optimizer-strategy=None $2 -o build_variants/lib_$1.so reinforcementlearning4meshgeneration_amd/csrc/meshenv_hip.hip
