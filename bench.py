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


def pmc_traffic(n_envs, workload="boundary0"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/): FETCH_SIZE and
    WRITE_SIZE are collected in separate --pmc runs, in KiB, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950.  None when no profile of this workload is committed (bench.py cannot run the profiler on itself)."""
    d, src = committed_profile(n_envs, workload)
    if d is None:
        return None, None, None
    t = d["hbm_traffic_bytes_per_launch"]
    # `traffic` is the guide's figure (2 x FETCH_SIZE + WRITE_SIZE: an upper bound for this kernel, whose loads mix 16-, 8- and
    # 4-byte-per-lane widths); the calibrated figure (FETCH_SIZE weighted by the factor measured for exactly this load mix,
    # tools/calib_fetch.sh) and the raw one are reported beside it, never instead of it
    return t["gfx950_corrected_(2*FETCH+WRITE)*1024"], src, {"calibrated": t.get("calibrated_(f*FETCH+WRITE)*1024"),
                                                             "fetch_calibration_factor": t.get("fetch_calibration_factor"),
                                                             "raw_FETCH+WRITE": t.get("raw_(FETCH+WRITE)*1024")}


def committed_profile(n_envs, workload):
    """Newest profiles/*_summary.json (tools/summarize_profile.py) taken on this workload, or (None, None)."""
    best = (None, None)
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_summary.json"):
                try:
                    d = json.load(open(os.path.join(pdir, f)))
                    cfg = d.get("bench_line_under_rocprof", {}).get("config", {})
                    if d.get("n_envs_per_gpu", cfg.get("n_envs_per_gpu")) == n_envs and d.get("workload_key", "boundary0") == workload:
                        best = (d, "profiles/" + f)
                except Exception:
                    pass
    return best


def instruction_side(n_envs, workload, launch_us):
    """The instruction-side view SURVEY 8(d) asks for next to the HBM fraction, from the committed SQ counter pass of this
    same command (bench.py cannot run the profiler on itself): VALU instructions per wave, the share of wave-cycles
    spent waiting / issuing VALU, and the launch's VALU issue floor -- every VALU instruction of the launch issued
    back to back at 4 cycles (one wave's issue cost, MI355X_MICROARCH.md) over the chip's 1024 SIMDs at 2.4 GHz."""
    d, src = committed_profile(n_envs, workload)
    if d is None:
        return None
    p = d.get("pmc_per_launch_mean", {})
    try:
        waves, valu, cyc = p["SQ_WAVES"], p["SQ_INSTS_VALU"], p["SQ_WAVE_CYCLES"]
        floor_us = valu * 4.0 / (1024 * 2.4e3)
        return {"source": src, "valu_insts_per_wave": valu / waves, "salu_insts_per_wave": p["SQ_INSTS_SALU"] / waves,
                "lds_insts_per_wave": p["SQ_INSTS_LDS"] / waves,
                "wait_frac": p["SQ_WAIT_ANY"] / cyc, "valu_active_frac": p["SQ_ACTIVE_INST_VALU"] / cyc,
                "issue_stall_frac": p["SQ_WAIT_INST_ANY"] / cyc,
                "valu_issue_floor_us": floor_us, "valu_issue_frac_of_launch": floor_us / launch_us if launch_us else None,
                "bound": "not HBM: the typical workgroup is bound by the dependent-issue latency of the ~12 % of waves that "
                         "extract an element (wait_frac), the launch's slowest workgroups (5-7 extractions on one CU) by "
                         "VALU issue (DESIGN.md section 5)"}
    except KeyError:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_quota():
    """CPUs this process may use: (cores, source) -- the cgroup CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us,
    rounded up) capped by the affinity mask; None when no quota is set."""
    def read(path):
        try:
            return open(path).read().split()
        except OSError:
            return None
    v2 = read("/sys/fs/cgroup/cpu.max")
    if v2 and len(v2) == 2 and v2[0] != "max":
        return -(-int(v2[0]) // int(v2[1])), "/sys/fs/cgroup/cpu.max = " + " ".join(v2)
    q, per = read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
    if q and per and int(q[0]) > 0:
        return -(-int(q[0]) // int(per[0])), f"cpu.cfs_quota_us / cpu.cfs_period_us = {q[0]} / {per[0]}"
    return None, ("/sys/fs/cgroup/cpu.max = " + " ".join(v2)) if v2 else "no cgroup CPU controller visible"


def cpu_baseline(n_envs, seed, doms, env_domain, wl, budget_s=12.0):
    """The CPU oracle (oracle/meshenv_ref.c, plain C, host libm) on the same workload, bounded sample."""
    from oracle.ref_lib import RefBatch, RefEnv, math_calls

    dom_of = env_domain if env_domain is not None else np.zeros(n_envs, np.int32)
    envs = [RefEnv.from_points(doms[int(dom_of[k])], cap_new=64) for k in range(n_envs)]
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
    math_calls(reset=True)
    dt1 = run(T1, 1)
    n_atan2, n_sin, n_cos = math_calls(reset=True)
    one = n_envs * T1 / dt1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # "all cores" = what this process may really use: min(cgroup CPU quota, affinity mask).  Without a quota the affinity
    # mask alone can overstate it (a one-GPU box of the pool shows 256 CPUs and gives 16), so 16 threads -- the pool's CPU
    # share per GPU -- are measured next to the full mask and the better run is reported.  The all-cores leg runs T vector
    # steps inside ONE OpenMP parallel region (meshenv_ref_rollout_batch: dynamic schedule over envs, chunks of 8 envs, each
    # thread stepping its envs T times) -- no fork / join per vector step, the CPU's best form of this workload.
    quota, quota_src = cpu_quota()
    want = int(os.environ.get("MESHENV_CPU_THREADS", "0"))
    if want > 0:
        tries = [want]
    elif quota is not None:
        tries = [max(1, min(quota, affinity))]
    else:
        tries = sorted({max(1, min(affinity, 16)), affinity})

    def run_region(T, threads):
        acts = rng.uniform(lo, hi, size=(T, n_envs, 3)).astype(np.float32)
        t0 = time.perf_counter()
        batch.rollout(acts, auto_reset=True, threads=threads)
        return time.perf_counter() - t0

    runs = []
    for cores in tries:
        dtc = run_region(4, cores)
        Tn = max(16, min(8000, int(0.5 * budget_s / len(tries) / (dtc / 4))))
        dtn = run_region(Tn, cores)
        runs.append(dict(value=n_envs * Tn / dtn, cores=cores, steps=Tn))
    best = max(runs, key=lambda r: r["value"])
    allc, cores, Tn = best["value"], best["cores"], best["steps"]
    per = float(n_envs * T1)
    return dict(value=one, unit="env-steps/s", cores=1, kind="port", cpu_model=cpu_model(),
                sample=f"{n_envs} envs on {wl} x {T1} uniform-random vector steps, oracle/meshenv_ref.c, 1 thread",
                all_cores=dict(value=allc, cores=cores, cgroup_cpu_quota=quota, cgroup_source=quota_src,
                               sched_getaffinity=affinity, cpu_count=os.cpu_count(), thread_counts_tried=runs,
                               openmp_schedule="one parallel region over the T steps of the sample; schedule(dynamic, 8) over envs",
                               sample=f"{n_envs} envs x {Tn} steps, OpenMP over envs, {cores} threads "
                                      f"(cgroup quota {quota}, sched_getaffinity allows {affinity}, the machine has {os.cpu_count()})"),
                fp64_transcendentals_per_env_step=dict(
                    atan2=n_atan2 / per, sin=n_sin / per, cos=n_cos / per,
                    note="libm calls of the reference algorithm (the oracle restates it call for call) on this sample; "
                         "the HIP kernels evaluate fewer (exact early rejects, cached observation of unchanged states)"))


def timing_group(K, time_every):
    """Launches per HIP-event bracket: the largest k <= time_every such that at least one bracket closes within K
    launches (meshenv_set_timing brackets launches [0, k), [2k, 3k), ... of the timed region)."""
    return max(1, min(int(time_every), int(K)))


def gather_steps(gather_every, K):
    """Steps per all-gather bucket: never more than the timed region holds, so a run of any length issues at least one
    collective inside the timed region (the driver's --steps 20 against the default bucket of 32)."""
    return max(1, min(int(gather_every), int(K)))


def run_region(xch, launch, t_first, count):
    """`count` vector steps, the exchange schedule included: step j of the region (absolute action index t_first + j)
    writes its message into slot j of the exchange, a full bucket goes out as one collective, the trailing partial bucket
    is flushed and everything in flight is waited for BEFORE the region ends -- every step's message is gathered inside
    the region that is timed.  Returns (collectives, step messages sent) of this region."""
    c0, s0 = (xch.collectives, xch.steps_sent) if xch is not None else (0, 0)
    for j in range(count):
        launch(t_first + j, xch.slot_ptr(j) if xch is not None else None)
        if xch is not None:
            xch.after_step(j)
    if xch is None:
        return 0, 0
    xch.flush()
    xch.drain()
    return xch.collectives - c0, xch.steps_sent - s0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU (weak scaling: every rank owns this many)")
    ap.add_argument("--envs-total", type=int, default=0,
                    help="strong scaling: this many environments in all, split evenly over the ranks (4096 -> 4096 / N per GPU); "
                         "overrides --envs, and the line says \"scaling\": \"strong\"")
    ap.add_argument("--log-capacity", type=int, default=0,
                    help="elements / created vertices logged per env and episode (generated_meshes, rl/boundary_env.py:192; "
                         "boundary.vertices, general/mesh.py:589-590).  0 = the env keeps counts only (the headline)")
    ap.add_argument("--workload", default="boundary0", choices=["boundary0", "d1", "mixed", "random"],
                    help="boundary0 = the headline config (BASELINE.json configs[1]); d1 = boundary16 (120-vertex ring, "
                         "configs[2] domain); mixed = d1/d2/d3 interleaved (configs[3] shape); random = one "
                         "GenerateRandomPolygon-style ring per env (configs[4] shape).  Domain coordinates of d1/d2/d3 "
                         "come from tests/golden (data recorded from the reference's ui/domains files)")
    ap.add_argument("--preroll", type=int, default=512,
                    help="untimed steps (fused rollouts) that bring the random policy to its steady state before the warm-up")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-every", type=int, default=32,
                    help="N > 1: steps per all-gather bucket (measured with a one-rank RCCL group: 8 -> 19.1, 32 -> 16.7, 64 -> 16.1 us per step against 15.4 without the exchange)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-process flow on a one-GPU box together with MESHENV_BENCH_DEVICE=0)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--clock-warmup-ms", type=float, default=250.0,
                    help="GPU busy time on a scratch handle before the warm-up steps (raises the clock from idle; 0 = off)")
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
    strong = args.envs_total > 0
    if strong:
        if args.envs_total % world:
            raise SystemExit(f"--envs-total {args.envs_total} is not a multiple of the {world} ranks")
        n = args.envs_total // world
    def golden_domain(name):
        tr = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        return [tuple(p) for p in tr["domain_xy"]]

    if args.workload == "boundary0":
        doms, env_domain, wl, wl_short = [boundary(0)], None, "boundary() (general/polygon.py:79-83, 30-vertex ring)", "boundary()"
    elif args.workload == "d1":
        doms, env_domain, wl, wl_short = [golden_domain("boundary16_biased_s2")], None, "ui/domains/boundary16.json (d1, 120-vertex ring)", "boundary16.json (d1)"
    elif args.workload == "mixed":
        doms = [golden_domain(x) for x in ("boundary16_biased_s2", "boundary15_biased_s5", "test1_biased_s42")]
        env_domain, wl, wl_short = np.arange(n, dtype=np.int32) % 3, "d1/d2/d3 interleaved (120/196/272-vertex rings)", "mixed d1/d2/d3"
    else:
        doms = None   # generated on the device (meshenv_create_random); fetched back only for the CPU baseline leg
        env_domain, wl, wl_short = np.arange(n, dtype=np.int32), "one GenerateRandomPolygon ring per env (ui/GenerateRandomPolygon.py:5-49 from random.Random(seed + k), clockwise, densified to 0.45, generated on the device)", "GenerateRandomPolygon"
    if doms is None:
        env = MeshVecEnv.from_random(n, 1000 + rank * n, device=local_rank, auto_reset=True, log_capacity=args.log_capacity)
    else:
        env = MeshVecEnv(doms, n_envs=n, env_domain=env_domain, device=local_rank, auto_reset=True, log_capacity=args.log_capacity)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    lo = torch.tensor([-1.0, -1.5, 0.0], device=dev)
    hi = torch.tensor([1.0, 1.5, 1.5], device=dev)
    actions = (lo + (hi - lo) * torch.rand((K + W, n, 3), device=dev, generator=gen)).to(torch.float32).contiguous()

    # packed exchange message for N > 1: [obs(18) | reward | done | complete] per env, float32
    from reinforcementlearning4meshgeneration_amd import sharding
    do_gather = world > 1 or os.environ.get("MESHENV_BENCH_FORCE_GATHER") == "1"
    GS = gather_steps(args.gather_every, args.steps)
    coll = "RCCL" if args.backend == "nccl" else args.backend
    xch = None
    if do_gather:
        # The step kernel writes the [n, 21] message of step t into slot t % GS of a [GS, n, 21] bucket; one
        # all_gather_into_tensor per bucket (RCCL, its own stream) overlaps the next bucket's steps -- actors run their
        # own policy copy, the gathered transitions feed the learner's replay buffer.  Two buckets, so a
        # bucket is only rewritten after its collective has completed.  One collective per GS steps keeps the
        # ~25 us host cost of a collective call off the per-step path (sharding.BucketExchange, covered on CPU by
        # tests/test_distributed_cpu.py).
        xch = sharding.BucketExchange(dist, torch, n, GS, dev, stage_to_cpu=args.backend != "nccl")

    # The timed loop calls the C-ABI entry point directly (the buffers are bound once): MeshVecEnv.step() adds argument
    # checks and a torch stream lookup per call, microseconds of host time that a 16 us launch cannot hide.
    env._bind_stream()
    L, handle = env._L, env._handle
    p_obs, p_rew, p_done, p_comp, p_term = (env.obs.data_ptr(), env.reward.data_ptr(), env.done.data_ptr(),
                                            env.complete.data_ptr(), env.terminal_obs.data_ptr())
    a_ptr, a_stride = actions.data_ptr(), n * 3 * 4

    def launch(t, slot_ptr):
        if slot_ptr is not None:
            L.meshenv_set_packed_output(handle, slot_ptr)
        rc = L.meshenv_step(handle, a_ptr + t * a_stride, p_obs, p_rew, p_done, p_comp, p_term, 1)
        if rc != 0:
            env._check(rc, "meshenv_step")

    # Clock warm-up (not steps of the measured environments): an idle MI355X sits at its lowest clock and a K = 20 timed
    # region lasts 0.3 ms -- too short for the clock to rise.  A scratch handle runs fused rollouts for ~0.25 s first.
    if args.clock_warmup_ms > 0:
        scratch = MeshVecEnv([boundary(0)], n_envs=4096, device=local_rank, auto_reset=True, log_capacity=0)
        sa = actions[:min(32, K + W), :min(n, 4096)].contiguous()
        if sa.shape[1] < 4096:
            sa = sa.repeat(1, (4096 + sa.shape[1] - 1) // sa.shape[1], 1)[:, :4096].contiguous()
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < args.clock_warmup_ms:
            for _ in range(8):
                scratch.rollout(sa)
            torch.cuda.synchronize()
        scratch.close()

    # Pre-roll (untimed, before the W warm-up steps): the uniform-random policy is brought to its stationary mix of ring
    # lengths / episode phases with fused rollouts on its own action stream, so that the measured workload does not depend
    # on how few warm-up steps the caller asks for (fresh envs all sit on the full 30-vertex ring: 15 % valid actions
    # against 11.7 % in the steady state).
    if args.preroll > 0:
        pr = (lo + (hi - lo) * torch.rand((64, n, 3), device=dev, generator=gen)).to(torch.float32).contiguous()
        for _ in range((args.preroll + 63) // 64):
            env.rollout(pr)
            pr = pr.roll(1, dims=1).contiguous()   # a different env/action pairing every round
        torch.cuda.synchronize()
        env._bind_stream()

    run_region(xch, launch, 0, W)
    torch.cuda.synchronize()
    c0 = env.counters()
    # HIP events bracket groups of k consecutive launches on the launch stream.  k never exceeds K, so at least one
    # group closes for every K >= 1 (a driver run with --steps 20 gets one group of 20, not an empty record).
    time_k = timing_group(K, args.time_every)
    if not args.no_kernel_timing:
        env.set_timing(time_k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_coll, n_sent = run_region(xch, launch, W, K)
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
    if do_gather:   # the exchange the workload names really ran inside the timed region, for every step of it
        assert n_coll >= 1 and n_sent == K, (n_coll, n_sent, K, GS)
    total_steps = K * n * world
    value = total_steps / elapsed
    out = {
        "metric": (f"env-steps/sec at N_envs={n * world} in all ({n} per GPU), {wl_short} domain, {world} MI355X" if strong else
                   f"env-steps/sec at N_envs={n} per GPU, {wl_short} domain, {world} MI355X"),
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n} vectorised envs per GPU on {wl}, "
                               "uniform-random float32 actions resident in HBM, auto-reset, one meshenv_step launch "
                               "per vector step, " + (f"log_capacity={args.log_capacity} (every accepted element and created vertex is "
                                                     "appended to the env's episode log, as generated_meshes / boundary.vertices are)"
                                                     if args.log_capacity > 0 else
                                                     "log_capacity=0 (element / vertex logs off: counts only)") + (f", + one async {coll} all-gather per {GS} steps of the [{GS},n,21] f32 obs/reward/done bucket written by the kernel ({n_coll} collectives covering all {n_sent} steps inside the timed region)" if do_gather else ""),
                   "n_envs_per_gpu": n, "n_envs_total": n * world, "parallelism": f"env-shard x{world}",
                   "log_capacity": args.log_capacity,
                   "scaling_mode": ("strong: --envs-total %d split over %d ranks" % (args.envs_total, world)) if strong else
                                   "weak: --envs per rank",
                   "gather_every": GS if do_gather else None, "collectives": n_coll if do_gather else 0,
                   "clock_warmup_ms": args.clock_warmup_ms, "preroll_steps": args.preroll,
                   "valid_action_rate": d["valid"] / max(1, d["steps"]), "mean_ring_len": d["sum_ring"] / max(1, d["steps"])},
    }
    # roofline of the dominant kernel (the one-step kernel): never omitted -- without a closed event group (timing
    # switched off) the per-launch time falls back to the wall clock of the timed region
    alg = algorithmic_bytes(d["steps"], d["valid"], d["sum_ring"], d["sum_ring_valid"]) / K
    if kt is not None and len(kt):
        avg_ms, min_ms, n_timed = float(np.mean(kt)), float(np.min(kt)), int(len(kt)) * time_k
        timing = "HIP events on the launch stream bracketing groups of %d consecutive launches" % time_k
    else:
        avg_ms = min_ms = 1e3 * elapsed / K
        n_timed, timing = K, "wall (no HIP-event group recorded): elapsed / steps, includes host launch gaps"
    achieved = alg / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src, traffic_alt = pmc_traffic(n, args.workload)
    # "bound" names the roofline the fraction is priced against (the contract's hbm | mfma; this path has no contraction);
    # "limited_by" says what the counters show the kernel is actually held by
    out["roofline"] = {"bound": "hbm", "limited_by": "instruction issue / dependent-issue latency, not HBM (instruction_side)",
                       "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                       "traffic_over_algorithmic": (traffic / alg) if traffic else None, "traffic_other_readings": traffic_alt,
                       "kernel": env.step_kernel,
                       "kernel_avg_us": avg_ms * 1e3, "kernel_min_us": min_ms * 1e3,
                       "algorithmic_bytes_per_launch": alg, "launches_timed": n_timed, "timing": timing,
                       "instruction_side": instruction_side(n, args.workload, avg_ms * 1e3)}
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
    env_doms = env.domains if (doms is None and rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    env.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, 99, doms if doms is not None else env_doms, env_domain, wl)
        out["cpu_baseline"]["gpu_over_cpu_all_cores"] = value / out["cpu_baseline"]["all_cores"]["value"]
        # vs_baseline stays null: BASELINE.md holds no published number for this metric (BASELINE.json "published": {}).
        # The ratio to the CPU baseline measured in this same run is reported under its own key.
        out["vs_cpu_baseline"] = {"all_cores": out["cpu_baseline"]["gpu_over_cpu_all_cores"],
                                  "one_core": value / out["cpu_baseline"]["value"]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if pg_up:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
