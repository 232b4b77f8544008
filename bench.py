#!/usr/bin/env python
"""bench.py -- env-steps/sec of the vectorised BoudaryEnv hot path on MI355X.

Workload (BASELINE.json configs[1]): 4096 independent boundary() environments per GPU, uniform-random
policy (actions pre-generated and resident in HBM), auto-reset on.  A "step" is one vector step = one
meshenv_step kernel launch over all 4096 envs.  For N > 1 every rank owns 4096 envs (weak scaling) and each
step ends with one RCCL all-gather of the packed (obs | reward | done | complete) message, the only exchange
the path has (a central learner needs the gathered observation batch).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def algorithmic_bytes(steps, valid, sum_ring, sum_ring_valid):
    """every step reads ring coords (16n) + candidate keys (8n) + stamps/flags (4n) and
    moves 158 B of scalars/action/obs/reward/flags; a valid extraction rewrites the ring (28n) + 48 B."""
    return 28 * sum_ring + 158 * steps + 28 * sum_ring_valid + 48 * valid


def pmc_traffic(n_envs):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/): FETCH_SIZE and
    WRITE_SIZE are collected in separate --pmc runs, in KiB, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950.  None when no profile of this workload is committed (bench.py cannot run the profiler on itself)."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_summary.json"):
                try:
                    d = json.load(open(os.path.join(pdir, f)))
                    if d.get("n_envs_per_gpu", d.get("bench_line_under_rocprof", {}).get("config", {}).get("n_envs_per_gpu")) == n_envs:
                        best = (d["hbm_traffic_bytes_per_launch"]["gfx950_corrected_(2*FETCH+WRITE)*1024"], "profiles/" + f)
                except Exception:
                    pass
    return best if best else (None, None)


def cpu_baseline(n_envs, seed, budget_s=12.0):
    """The CPU oracle (oracle/meshenv_ref.c, plain C, host libm) on the same workload, bounded sample."""
    from oracle.ref_lib import RefBatch, RefEnv
    from reinforcementlearning4meshgeneration_amd.domains import boundary

    envs = [RefEnv.from_points(boundary(0), cap_new=64) for _ in range(n_envs)]
    batch = RefBatch(envs)
    batch.reset()
    rng = np.random.default_rng(seed)
    lo, hi = np.array([-1, -1.5, 0.0]), np.array([1, 1.5, 1.5])

    def run(T, threads):
        acts = rng.uniform(lo, hi, size=(T, n_envs, 3)).astype(np.float32)
        t0 = time.perf_counter()
        for t in range(T):
            batch.step(acts[t], auto_reset=True, threads=threads)
        return time.perf_counter() - t0

    dt = run(4, 1)  # calibrate
    T1 = max(8, min(2000, int(0.5 * budget_s / (dt / 4))))
    dt1 = run(T1, 1)
    one = n_envs * T1 / dt1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("MESHENV_CPU_THREADS", "16"))))
    dtc = run(2, cores)
    Tn = max(8, min(4000, int(0.5 * budget_s / (dtc / 2))))
    dtn = run(Tn, cores)
    allc = n_envs * Tn / dtn
    return dict(value=one, unit="env-steps/s", cores=1, kind="port",
                sample=f"{n_envs} boundary() envs x {T1} uniform-random vector steps, oracle/meshenv_ref.c, 1 thread",
                all_cores=dict(value=allc, cores=cores, sample=f"{n_envs} envs x {Tn} steps, OpenMP over envs"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--workload", default="boundary0", choices=["boundary0", "d1", "mixed", "random"],
                    help="boundary0 = the headline config (BASELINE.json configs[1]); d1 = boundary16 (120-vertex ring, "
                         "configs[2] domain); mixed = d1/d2/d3 interleaved (configs[3] shape); random = one "
                         "GenerateRandomPolygon-style ring per env (configs[4] shape).  Domain coordinates of d1/d2/d3 "
                         "come from tests/golden (data recorded from the reference's ui/domains files)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-every", type=int, default=8, help="N > 1: steps per all-gather bucket")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-process flow on a one-GPU box together with MESHENV_BENCH_DEVICE=0)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-every", type=int, default=25,
                    help="HIP events bracket every other group of k consecutive launches of the timed region (an event pair costs ~8 us of stream time, so k is kept large)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from reinforcementlearning4meshgeneration_amd.domains import boundary
    from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pg_up = False
    if "MESHENV_BENCH_DEVICE" in os.environ:  # rehearsal: several ranks on one GPU (gloo backend only)
        local_rank = int(os.environ["MESHENV_BENCH_DEVICE"])
    if world > 1 or os.environ.get("MESHENV_BENCH_FORCE_GATHER") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        pg_up = True
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    n, K, W = args.envs, args.steps, args.warmup
    def golden_domain(name):
        tr = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        return [tuple(p) for p in tr["domain_xy"]]

    if args.workload == "boundary0":
        doms, env_domain, wl = [boundary(0)], None, "boundary() (general/polygon.py:79-83, 30-vertex ring)"
    elif args.workload == "d1":
        doms, env_domain, wl = [golden_domain("boundary16_biased_s2")], None, "ui/domains/boundary16.json (d1, 120-vertex ring)"
    elif args.workload == "mixed":
        doms = [golden_domain(x) for x in ("boundary16_biased_s2", "boundary15_biased_s5", "test1_biased_s42")]
        env_domain, wl = np.arange(n, dtype=np.int32) % 3, "d1/d2/d3 interleaved (120/196/272-vertex rings)"
    else:
        from reinforcementlearning4meshgeneration_amd.domains import random_domain
        doms = [random_domain(1000 + rank * n + k) for k in range(n)]
        env_domain, wl = np.arange(n, dtype=np.int32), "one random star-shaped ring per env (GenerateRandomPolygon restated, densified)"
    env = MeshVecEnv(doms, n_envs=n, env_domain=env_domain, device=local_rank, auto_reset=True, log_capacity=0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    lo = torch.tensor([-1.0, -1.5, 0.0], device=dev)
    hi = torch.tensor([1.0, 1.5, 1.5], device=dev)
    actions = (lo + (hi - lo) * torch.rand((K + W, n, 3), device=dev, generator=gen)).to(torch.float32).contiguous()

    # packed exchange message for N > 1: [obs(18) | reward | done | complete] per env, float32
    from reinforcementlearning4meshgeneration_amd import sharding
    do_gather = world > 1 or os.environ.get("MESHENV_BENCH_FORCE_GATHER") == "1"
    GS = max(1, args.gather_every)
    if do_gather:
        # The step kernel writes the [n, 21] message of step t into slot t % GS of a [GS, n, 21] bucket; one
        # all_gather_into_tensor per bucket (RCCL, its own stream) overlaps the next bucket's steps -- actors run their
        # own policy copy, the gathered transitions feed the learner's replay buffer.  Two buckets, so a
        # bucket is only rewritten after its collective has completed.  One collective per GS steps keeps the
        # ~25 us host cost of a collective call off the per-step path.
        on_dev = args.backend == "nccl"
        buckets = [torch.zeros((GS, n, sharding.MSG_DIM), dtype=torch.float32, device=dev) for _ in range(2)]
        gdev = dev if on_dev else torch.device("cpu")
        gath = [torch.empty((world * GS, n, sharding.MSG_DIM), dtype=torch.float32, device=gdev) for _ in range(2)]  # rank-major
        works = [None, None]

    def one_step(t):
        if do_gather:
            slot, b = t % GS, (t // GS) & 1
            if slot == 0 and works[b] is not None:
                works[b].wait()          # stream-level wait for the collective that last used this bucket
                works[b] = None
            env.set_packed_output(buckets[b][slot])
        env.step(actions[t])
        if do_gather and slot == GS - 1:
            if on_dev:
                works[b] = dist.all_gather_into_tensor(gath[b], buckets[b], async_op=True)
            else:
                dist.all_gather_into_tensor(gath[b], buckets[b].cpu())

    def drain():
        if do_gather:
            for b in (0, 1):
                if works[b] is not None:
                    works[b].wait()
                    works[b] = None

    for t in range(W):
        one_step(t)
    drain()
    torch.cuda.synchronize()
    c0 = env.counters()
    if not args.no_kernel_timing:
        env.set_timing(args.time_every)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        one_step(t)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kt = None
    if not args.no_kernel_timing:
        kt = env.kernel_times_ms()
        env.set_timing(0)
    c1 = env.counters()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    d = {k: c1[k] - c0[k] for k in c0}
    assert d["steps"] == K * n, (d, K, n)
    total_steps = K * n * world
    value = total_steps / elapsed
    out = {
        "metric": "env-steps/sec at N_envs=4096 per GPU, boundary() domain",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n} vectorised envs per GPU on {wl}, "
                               "uniform-random float32 actions resident in HBM, auto-reset, one meshenv_step launch "
                               "per vector step" + (f", + one async RCCL all-gather per {GS} steps of the [{GS},n,21] f32 obs/reward/done bucket written by the kernel" if world > 1 else ""),
                   "n_envs_per_gpu": n, "n_envs_total": n * world, "parallelism": f"env-shard x{world}",
                   "valid_action_rate": d["valid"] / max(1, d["steps"]), "mean_ring_len": d["sum_ring"] / max(1, d["steps"])},
    }
    if kt is not None and len(kt):
        alg = algorithmic_bytes(d["steps"], d["valid"], d["sum_ring"], d["sum_ring_valid"]) / K
        avg_ms = float(np.mean(kt))
        achieved = alg / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(n)
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                           "kernel": ("meshenv::k_step_group<%d, true>" % env.group_size) if env.group_size > 1 else "meshenv::k_step<false, true>",
                           "kernel_avg_us": avg_ms * 1e3, "kernel_min_us": float(np.min(kt)) * 1e3,
                           "algorithmic_bytes_per_launch": alg, "launches_timed": int(len(kt)) * args.time_every,
                           "timing": "HIP events on the launch stream bracketing groups of %d consecutive launches" % args.time_every}
    if world == 1 and args.workload == "boundary0":
        # informational: the same steps fused T per launch (meshenv_rollout, open-loop actions); never part of `value`
        Tr = min(64, K + W)
        env.rollout(actions[:Tr])
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        for _ in range(3):
            env.rollout(actions[:Tr])
        torch.cuda.synchronize()
        out["fused_rollout"] = {"value": 3 * Tr * n / (time.perf_counter() - tr0), "unit": "env-steps/s", "steps_per_launch": Tr,
                                "note": "meshenv_rollout: T consecutive steps per kernel launch, open-loop only"}
    env.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, seed=99)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if pg_up:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
