"""Dev tool: what would a PERFECT dealing of valid extractions over the CUs be worth?  

All 4096 envs sit on the freshly reset boundary() ring, so whether an action extracts an element depends on the action
alone.  The same 4096 actions are then laid out three ways over the 256 workgroups of k_step_group<16> (one per CU):
  random    -- as drawn: the number of valid extractions per workgroup is binomial (what the bench runs),
  balanced  -- every workgroup gets the same number (+-1): the best any cross-CU dealing could reach, at ZERO transfer cost,
  packed    -- all valid actions in the first workgroups (16 per CU): the worst case.
One launch each (reset before it, HIP events around it), median of many repetitions.  random - balanced is the most a
cross-CU hand-over of pending extractions could gain before paying for the hand-over itself.
"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
from reinforcementlearning4meshgeneration_amd.domains import boundary

n, G, reps = 4096, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 300
env = MeshVecEnv([boundary(0)], n_envs=n, auto_reset=False)
g = torch.Generator(device="cuda"); g.manual_seed(3)
lo = torch.tensor([-1., -1.5, 0.], device="cuda"); hi = torch.tensor([1., 1.5, 1.5], device="cuda")
a = (lo + (hi - lo) * torch.rand((n, 3), device="cuda", generator=g)).float().contiguous()
env.reset()
_, rew, _, _ = env.step(a)
valid = (rew != -1.0).cpu().numpy()
nv = int(valid.sum())
print(f"{nv} of {n} actions extract an element on the reset ring ({100.0 * nv / n:.1f} %), kernel {env.step_kernel}")
vi, ii = np.nonzero(valid)[0], np.nonzero(~valid)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0   # e.g. 0.113: the steady-state valid rate of the random policy
if frac > 0:   # thin the valid actions out to that rate (replace the surplus by copies of invalid actions)
    keep = int(round(frac * n))
    drop = vi[keep:]
    a[torch.from_numpy(drop).cuda()] = a[torch.from_numpy(ii[:len(drop)]).cuda()]
    valid[drop] = False
    vi, ii = np.nonzero(valid)[0], np.nonzero(~valid)[0]
    print(f"thinned to {len(vi)} valid actions ({100.0 * len(vi) / n:.1f} %)")

def layout(kind):
    order = np.empty(n, np.int64)
    if kind == "random":
        order[:] = np.random.default_rng(1).permutation(n)
    elif kind == "packed":
        order[:] = np.concatenate([vi, ii])
    else:  # balanced: deal the valid ones round-robin over the workgroups, fill the rest with invalid ones
        wgs = n // G
        slots = [[] for _ in range(wgs)]
        for k, e in enumerate(vi):
            slots[k % wgs].append(e)
        it = iter(ii)
        for s in slots:
            while len(s) < G:
                s.append(next(it))
        order[:] = np.concatenate([np.array(s) for s in slots])
    return a[torch.from_numpy(order).cuda()].contiguous(), np.bincount(np.arange(n)[valid[order]] // G, minlength=n // G)

# clock warm-up
for _ in range(300):
    env.reset(); env.step(a)
torch.cuda.synchronize()
res = {}
for kind in ("random", "balanced", "packed", "random", "balanced"):
    acts, per_wg = layout(kind)
    ts = []
    for r in range(reps):
        env.reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step(acts); e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = np.array(ts)
    res.setdefault(kind, []).append(np.median(ts))
    print(f"{kind:9s}: valid per workgroup max {per_wg.max()} mean {per_wg.mean():.2f} | step launch median {np.median(ts):.2f} us  p10 {np.percentile(ts, 10):.2f}  p90 {np.percentile(ts, 90):.2f}")
print({k: [round(float(x), 2) for x in v] for k, v in res.items()})
