"""GPU: the diagnostic / fallback builds of the library stay correct.
  -DMESHENV_NO_HELPER   CU-group kernel with the reward on the update wave (no helper wavefront, no inter-wave LDS flags)
  -DMESHENV_NO_FILTERS  kernels without the exactness-preserving shortcuts (reference evaluation everywhere: the exact
                        atan2 instead of cw_fast, no collinearity shortcut, ...)
  -DMESHENV_DEV         the measured-slower experiments the shipped library no longer instantiates: the speculative CU-group
                        kernel (k_step_spec, MESHENV_SPEC=1), the T-steps-per-launch closed loop (k_step_group_actor_T) and
                        k_step_group<4> -- their own tests (test_gpu_parity.py, test_gpu_actor.py) re-run against that build
Each variant is compiled here with hipcc (the GPU box has the same image) and run in a process of its own (MESHENV_LIB
selects the library at load time) through a short lockstep with the oracle; the default library's outputs on the same
actions must be identical to the variants' bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import json, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from lockstep import run_lockstep
from reinforcementlearning4meshgeneration_amd import MeshVecEnv, boundary
n, T = 4096, 64
rng = np.random.default_rng(31)
a = rng.uniform([-1, -1.5, 0], [1, 1.5, 1.5], size=(T, n, 3))
pick = rng.random((T, n)) < 0.5
b = np.stack([rng.uniform(-1, 1, (T, n)), rng.uniform(0.2, 1.0, (T, n)), rng.uniform(0.3, 1.2, (T, n))], axis=2)
a[pick] = b[pick]; a = a.astype(np.float32)
st = run_lockstep(torch, [boundary(0)], np.zeros(n, np.int32), a, check_every=32, sample=64)
env = MeshVecEnv([boundary(0)], n_envs=n); env.reset()
acts = torch.from_numpy(a).cuda(); h = 0.0
for t in range(T):
    o, r, d, c = env.step(acts[t]); h += float(o.double().sum()) + float(r.sum()) + float(d.sum())
kernel0 = env.step_kernel
env.close()
# long rings (d1 / d2 / d3: 120 / 196 / 272 vertices): the multi-chunk pre-filters of point_inside and of the observation scan
# against the unfiltered forms of -DMESHENV_NO_FILTERS -- CU-group kernel with ragged LDS (4096 envs) and k_step (600 envs)
import os
gold = os.path.join(sys.argv[1], "tests", "golden")
doms = [[tuple(p) for p in np.load(os.path.join(gold, f + ".npz"))["domain_xy"]] for f in ("boundary16_biased_s2", "boundary15_biased_s5", "test1_biased_s42")]
h_long, valid_long, kernels = 0.0, 0, []
for nl in (4096, 600):
    el = MeshVecEnv(doms, env_domain=(np.arange(nl) % 3).astype(np.int32)); el.reset()
    kernels.append(el.step_kernel)
    rl = np.random.default_rng(77)
    for t in range(160):
        al = rl.uniform([-1, -1.5, 0], [1, 1.5, 1.5], size=(nl, 3))
        pk = rl.random(nl) < 0.6
        bl = np.stack([rl.uniform(-1, 1, nl), rl.uniform(0.2, 1.0, nl), rl.uniform(0.3, 1.2, nl)], axis=1)
        al[pk] = bl[pk]
        o, r, d, c = el.step(torch.from_numpy(al.astype(np.float32)).cuda())
        h_long += float(o.double().sum()) + float(r.sum()) + float(d.sum())
    valid_long += el.counters()["valid"]
    el.close()
print(json.dumps(dict(valid=st["valid"], obs_mismatch=st["obs_mismatch"], kernel=kernel0, checksum=h,
                      checksum_long=h_long, valid_long=valid_long, kernels_long=kernels)))
"""


def _build(name, flags):
    out = os.path.join(ROOT, "build_variants", f"lib_{name}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    from reinforcementlearning4meshgeneration_amd.build import CSRC, HIPCC_FLAGS, hipcc_path
    subprocess.check_call([hipcc_path(), *HIPCC_FLAGS, *flags, "-o", out, os.path.join(CSRC, "meshenv_hip.hip")], cwd=CSRC)
    return out


def _run(lib):
    env = dict(os.environ)
    if lib:
        env["MESHENV_LIB"] = lib
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_no_helper_and_no_filters_builds_agree_with_the_default_library():
    base = _run(None)
    assert base["kernel"] == "meshenv::k_step_group<16, true, false, true>" and base["valid"] > 0.1 * 4096 * 64
    for name, flags in (("nohelper", ["-DMESHENV_NO_HELPER"]), ("nofilters", ["-DMESHENV_NO_FILTERS"])):
        res = _run(_build(name, flags))
        assert res["valid"] == base["valid"] and res["obs_mismatch"] == 0, (name, res)
        assert res["checksum"] == base["checksum"], (name, res["checksum"], base["checksum"])
        # long rings: the pre-filtered passes change nothing, to the last bit of every observation and reward
        assert res["valid_long"] == base["valid_long"] > 20000 and res["checksum_long"] == base["checksum_long"], (name, res, base)
    assert base["kernels_long"] == ["meshenv::k_step_group<16, true, true, false>", "meshenv::k_step<false, true, false, false, false>"]


def test_dev_build_runs_the_experimental_kernels():
    """The kernels behind -DMESHENV_DEV keep their parity tests: the speculative kernel against the oracle and the default
    kernel, the T-step closed loop against T single-step launches (bit-identical histories)."""
    lib = _build("dev", ["-DMESHENV_DEV"])
    env = dict(os.environ, MESHENV_LIB=lib)
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py") + "::test_speculative_group_kernel_variant",
                        os.path.join(ROOT, "tests", "test_gpu_actor.py") + "::test_t_steps_per_launch_equals_single_step_launches"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0 and "2 passed" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
