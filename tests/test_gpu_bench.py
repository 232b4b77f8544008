"""GPU: bench.py as the driver runs it, in child processes of its own (a process group must be initialised before the first
GPU call of the process that owns it).

* the one-rank `nccl` group: `MESHENV_BENCH_FORCE_GATHER=1 bench.py --gpus 1 --steps 20 --warmup 5` puts the N > 1 exchange
  (sharding.BucketExchange: the step kernel writes the [GS, n, 21] bucket, one async RCCL all_gather_into_tensor per bucket on
  device tensors, flush + drain inside the timed region) under a driver-run test -- the same code an 8-GPU node runs, with a
  world of one; the line is kept under gpurun_out/ (copied to profiles/ by hand);
* the contract keys the driver parses, the log_capacity / scaling fields the line states, and the cgroup-aware CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, extra_env=None, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_one_rank_rccl_group_gathers_every_step_inside_the_timed_region():
    base = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    line = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], {"MESHENV_BENCH_FORCE_GATHER": "1"})
    cfg = line["config"]
    assert base["config"]["collectives"] == 0 and base["config"]["gather_every"] is None
    assert cfg["collectives"] >= 1 and cfg["gather_every"] == 20 and "RCCL all-gather" in cfg["workload"]
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["scaling"] == "weak"
    # the exchange costs little: the collective overlaps the next bucket's steps (one call per 20 steps)
    assert line["value"] > 0.75 * base["value"], (line["value"], base["value"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"no_gather": base, "one_rank_rccl_gather": line}, open(os.path.join(ROOT, "gpurun_out", "rccl_one_rank.json"), "w"), indent=1)


def test_bench_line_states_log_capacity_scaling_and_cpu_quota():
    line = _bench(["--steps", "20", "--warmup", "5", "--log-capacity", "64", "--envs-total", "4096"],
                  {"MESHENV_CPU_THREADS": "0"})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["scaling"] == "strong" and line["config"]["n_envs_per_gpu"] == 4096 and line["config"]["log_capacity"] == 64
    assert "log_capacity=64" in line["config"]["workload"] and "N_envs=4096 in all" in line["metric"]
    allc = line["cpu_baseline"]["all_cores"]
    assert allc["cores"] >= 1 and "cgroup_cpu_quota" in allc and "openmp_schedule" in allc
    if allc["cgroup_cpu_quota"] is not None:
        assert allc["cores"] == min(allc["cgroup_cpu_quota"], allc["sched_getaffinity"])
    r = line["roofline"]
    assert r["kernel"] == "meshenv::k_step_group<16, true, false, true>" and 5 < r["kernel_avg_us"] < 40 and 0 < r["frac"] < 1
