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


def test_two_rank_launch_as_the_driver_starts_it_gloo_rehearsal_on_one_gpu():
    """The N > 1 launch path exactly as the driver starts it -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps 20 --warmup 5` -- rehearsed on the one GPU of this box:
    both ranks share device 0 (MESHENV_BENCH_DEVICE=0) and the exchange goes over gloo (RCCL refuses two ranks on one device;
    the one-rank test above covers the `nccl` backend).  Checks the rank plumbing (RANK / LOCAL_RANK / WORLD_SIZE), the env
    sharding, the bucket exchange across two processes, the max-over-ranks timing and that rank 0 alone prints the line."""
    env = dict(os.environ, MESHENV_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29583", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--backend", "gloo", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 only
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and cfg["n_envs_per_gpu"] == 4096 and cfg["n_envs_total"] == 8192
    assert cfg["collectives"] >= 1 and cfg["gather_every"] == 20 and "gloo all-gather" in cfg["workload"]
    # (gloo stages every bucket through host memory and both ranks share one GPU: a plumbing rehearsal, not a rate)
    assert line["value"] > 1e6 and abs(line["value"] - 8192 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6
    # strong scaling: the same total split over the ranks
    p = subprocess.run(cmd + ["--envs-total", "4096"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["config"]["n_envs_per_gpu"] == 2048 and line["config"]["n_envs_total"] == 4096
