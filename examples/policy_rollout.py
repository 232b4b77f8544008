#!/usr/bin/env python
"""BASELINE.json configs[2]: 4096 envs of ui/domains/boundary16.json (d1) on one MI355X with an SB3-SAC-shaped
policy in the loop, everything on the GPU (no numpy hop).

The actor is the reference's architecture (rl/baselines/RL_Mesh.py:183-196: MlpPolicy, ReLU, net_arch [128, 128, 128];
SAC's squashed-Gaussian actor: mean and log_std heads, tanh squash, rescale to the action Box), random-initialised with
seed 999 -- SB3 itself is not installed in this image and there is no checkpoint to load.

    python examples/policy_rollout.py [--envs 4096] [--steps 300] [--domain d1|boundary0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--domain", default="d1", choices=["d1", "boundary0"])
    ap.add_argument("--deterministic", action="store_true")
    ap.add_argument("--chunk", type=int, default=32, help="t-steps: vector steps per launch")
    ap.add_argument("--actor", default="fused", choices=["t-steps", "one-launch", "fused", "fused-ext-noise", "graph", "eager"],
                    help="t-steps = --chunk vector steps of the closed loop per launch (meshenv_step_actor_multi); "
                         "one-launch = env step + actor forward in ONE kernel per vector step (meshenv_step_actor); "
                         "fused = the hand-written HIP actor kernel, exploration noise drawn inside it (one launch); "
                         "fused-ext-noise = same kernel fed by torch's normal_() (two launches); graph = the torch MLP captured as one HIP "
                         "graph; eager = the torch MLP launch by launch")
    args = ap.parse_args()
    import torch

    from reinforcementlearning4meshgeneration_amd import MeshVecEnv, boundary
    from reinforcementlearning4meshgeneration_amd.vec_env import ACTION_HIGH, ACTION_LOW

    if args.domain == "d1":
        tr = np.load(os.path.join(ROOT, "tests", "golden", "boundary16_biased_s2.npz"))
        dom = [tuple(p) for p in tr["domain_xy"]]
    else:
        dom = boundary(0)
    dev = torch.device("cuda", 0)
    torch.manual_seed(999)
    trunk = torch.nn.Sequential(torch.nn.Linear(18, 128), torch.nn.ReLU(), torch.nn.Linear(128, 128), torch.nn.ReLU(),
                                torch.nn.Linear(128, 128), torch.nn.ReLU()).to(dev)
    mu_head, log_std_head = torch.nn.Linear(128, 3).to(dev), torch.nn.Linear(128, 3).to(dev)
    low = torch.as_tensor(ACTION_LOW, device=dev)
    high = torch.as_tensor(ACTION_HIGH, device=dev)

    @torch.no_grad()
    def act(obs):
        h = trunk(obs)
        mu = mu_head(h)
        if not args.deterministic:
            std = log_std_head(h).clamp(-20, 2).exp()
            mu = mu + std * torch.randn_like(mu)          # default generator: capturable in a HIP graph
        a = torch.tanh(mu)                                 # SB3 squashes, then unscale_action maps [-1, 1] -> Box
        return (low + 0.5 * (a + 1.0) * (high - low)).contiguous()

    env = MeshVecEnv([dom], n_envs=args.envs, device=0)
    obs = env.reset()          # env.obs: the kernel always writes observations into this tensor
    if args.actor.startswith("fused") or args.actor in ("one-launch", "t-steps"):
        from reinforcementlearning4meshgeneration_amd.actor import FusedActor
        fused = FusedActor.from_torch([trunk[0], trunk[2], trunk[4]], mu_head, log_std_head)
        actions = torch.empty((args.envs, 3), dtype=torch.float32, device=dev)
        noise = torch.empty((args.envs, 3), dtype=torch.float32, device=dev)
        draw = [0]

        def policy(o):
            if args.deterministic:
                return fused.forward(o, None, out=actions)
            if args.actor in ("fused", "one-launch", "t-steps"):
                draw[0] += 1
                return fused.sample(o, 999, draw[0], out=actions)
            return fused.forward(o, noise.normal_(), out=actions)
    elif args.actor == "graph":
        # ~25 tiny launches of the actor -> one graph replay; input = the env's observation buffer, output static
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                static_act = act(env.obs)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_act = act(env.obs)

        def policy(_obs):
            graph.replay()
            return static_act
    else:
        policy = act
    # timed loop: policy + step only (one, two or three launches per vector step)
    if args.actor == "t-steps":
        nxt = policy(obs)
        T = args.chunk
        args.warmup = (args.warmup + T - 1) // T * T
        args.steps = (args.steps + T - 1) // T * T
        for t in range(0, args.warmup + args.steps, T):
            if t == args.warmup:
                torch.cuda.synchronize()
                c0 = env.counters()
                t0 = time.perf_counter()
            hist = env.step_actor_T(fused, nxt, T, seed=999, counter=draw[0] + 1, sample=not args.deterministic)
            draw[0] += T
            nxt = hist["actions"][T]
        obs = env.obs
    elif args.actor == "one-launch":
        nxt = policy(obs)
        for t in range(args.warmup + args.steps):
            if t == args.warmup:
                torch.cuda.synchronize()
                c0 = env.counters()
                t0 = time.perf_counter()
            draw[0] += 1
            obs, rew, done, comp, nxt = env.step_actor(fused, nxt, seed=999, counter=draw[0], sample=not args.deterministic)
    else:
        for t in range(args.warmup + args.steps):
            if t == args.warmup:
                torch.cuda.synchronize()
                c0 = env.counters()
                t0 = time.perf_counter()
            obs, rew, done, comp = env.step(policy(obs))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = env.counters()
    steps = c1["steps"] - c0["steps"]
    # untimed: episode statistics over a few more steps (each reduction is a launch of its own)
    ep_done = ep_complete = 0
    rew_sum = 0.0
    for t in range(50):
        obs, rew, done, comp = env.step(policy(obs))
        ep_done += int(done.sum())
        ep_complete += int((done & comp).sum())
        rew_sum += float(rew.sum())
    print(json.dumps({"workload": f"{args.envs} envs, domain {args.domain} ({len(dom)}-vertex ring), SAC-shaped MLP actor 18-128-128-128-3 on the same GPU, actor = " + args.actor,
                      "env_steps_per_s": steps / dt, "us_per_vector_step": 1e6 * dt / args.steps,
                      "valid_action_rate": (c1["valid"] - c0["valid"]) / steps,
                      "episodes_finished_per_1000_env_steps": 1000.0 * ep_done / (50 * args.envs),
                      "completed_fraction": ep_complete / max(1, ep_done), "mean_reward": rew_sum / (50 * args.envs),
                      # for tools/profile_set.sh: algorithmic bytes per launch of the kernels of this loop (SURVEY 8d's step
                      # formula from the work counters; the actor reads 72 B and writes 12 B per env + its 141 KB of weights)
                      "profile_kernels": [
                          {"match": "k_step_group", "algorithmic_bytes_per_launch":
                              (28 * (c1["sum_ring"] - c0["sum_ring"]) + 158 * steps + 28 * (c1["sum_ring_valid"] - c0["sum_ring_valid"])
                               + 48 * (c1["valid"] - c0["valid"])) / args.steps + (84.0 * args.envs + 141e3 if args.actor in ("one-launch", "t-steps") else 0.0),
                           "note": "env step (28 sum_n + 158 steps + 28 sum_n_valid + 48 valid) per vector step; the one-launch kernel adds the actor's 84 B per env + weights"},
                          {"match": "k_actor_forward", "algorithmic_bytes_per_launch": 84.0 * args.envs + 141e3,
                           "note": "18 f32 observations in, 3 f32 actions out per env, 35 k f32 weights once"}]}))
    env.close()


if __name__ == "__main__":
    main()
