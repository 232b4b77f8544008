"""Builds libmeshenv_hip.so (gfx950) in-tree with hipcc.  `python -m reinforcementlearning4meshgeneration_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_NAME = "libmeshenv_hip.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
SOURCES = ["meshenv_hip.hip"]
ARCH = "gfx950"

# -ffp-contract=off: the reference is CPython float arithmetic, which never fuses a*b+c.
# -fhip-fp32-correctly-rounded-divide-sqrt: numpy's float32 round() divides in IEEE float32.
# -mllvm -disable-machine-licm: the kernels' loops run once or twice (64-vertex chunks of a ring); hoisting constant
# materialisation out of them only lengthens live ranges (k_step<false>: 94 -> 89 VGPRs, 86 -> 78 spilled SGPRs;
# measured +1..3 % in the throughput workloads, neutral on the headline: profiles/r03_ab_all.log).
# -mllvm -amdgpu-atomic-optimizer-strategy=None: the LDS atomics of the observation scan (find_next_state) have one to
# three active lanes; the optimiser's wave reduction in front of them is the very DPP sequence they replace
# (headline 14.58 -> 14.42 us, 65 536 envs +1 %, rollout +2 %; tools/ab_all.sh).
HIPCC_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
               "-fno-fast-math", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
               "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libmeshenv_hip.so)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    # every header of csrc/ is a dependency of the one translation unit (meshenv_hip.hip includes them all)
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip"))]
    deps.append(os.path.join(os.path.dirname(PKG_DIR), "include", "meshenv.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc_path(), *HIPCC_FLAGS, *extra_flags, "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    # what this binary was built from (tools/source_state.py: a profile refuses to describe a library that is not the sources')
    import hashlib
    import json
    h = hashlib.sha256()
    root = os.path.dirname(PKG_DIR)
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))) + [os.path.join(root, "include", "meshenv.h")]
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    json.dump({"source_sha256": h.hexdigest(), "flags": [*HIPCC_FLAGS, *extra_flags]}, open(LIB_PATH + ".source", "w"))
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
