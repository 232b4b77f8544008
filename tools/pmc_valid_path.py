"""Dev tool (run under rocprofv3 --pmc ...): instruction count of ONE valid rule-0 extraction in the CU-group kernel.
All 4096 envs on the freshly reset boundary() ring; launches alternate between A = every workgroup holds exactly one valid
rule-0 action (its other 15 envs take a rule-0 point far outside) and B = all 16 take the outside point.  Per-launch
counter(A) - counter(B), divided by 256 workgroups = what the valid action adds (its longer check, the update, the reward
helper) beyond a rejected one.  tools/pmc_valid_path.sh prints it.
usage: python tools/pmc_valid_path.py [pairs]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
from reinforcementlearning4meshgeneration_amd.domains import boundary
n, G = 4096, 16
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
env = MeshVecEnv([boundary(0)], n_envs=n, auto_reset=False)
g = torch.Generator(device="cuda"); g.manual_seed(3)
lo = torch.tensor([-0.4, -1.5, 0.], device="cuda"); hi = torch.tensor([0.4, 1.5, 1.5], device="cuda")
a = (lo + (hi - lo) * torch.rand((n, 3), device="cuda", generator=g)).float().contiguous()
env.reset()
_, rew, _, _ = env.step(a)
valid = np.nonzero((rew != -1.0).cpu().numpy())[0]
assert len(valid) >= n // G, len(valid)
outside = torch.tensor([0.0, -1.4, 1.4], device="cuda")
A = outside.repeat(n, 1).contiguous()
A[::G] = a[torch.from_numpy(valid[:n // G]).cuda()]
B = outside.repeat(n, 1).contiguous()
for k in range(pairs):
    env.reset(); o, r, d, c = env.step(A)
    nva = int((r != -1.0).sum())
    env.reset(); o, r, d, c = env.step(B)
    nvb = int((r != -1.0).sum())
torch.cuda.synchronize()
print("valid per launch A / B:", nva, nvb)
