"""Dev tool (run under rocprofv3 --pmc ...): launches of one action pattern, to count instructions per code path.
usage: python tools/pmc_paths.py {memo|rule0_out|uniform} [steps]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinforcementlearning4meshgeneration_amd import MeshVecEnv, boundary
mode = sys.argv[1]; T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = 4096
env = MeshVecEnv([boundary(0)], n_envs=n); env.reset()
rng = np.random.default_rng(0)
if mode == "memo":          # rule +1 on the reset state every step: first launch evaluates the quad, the rest hit the memo (or are valid)
    a = np.tile(np.array([1.0, 0.0, 0.0], np.float32), (T, n, 1))
elif mode == "rule0_out":   # rule 0 with a point far outside: point-in-polygon fails
    a = np.tile(np.array([0.0, -1.4, 1.4], np.float32), (T, n, 1))
else:
    a = rng.uniform([-1, -1.5, 0], [1, 1.5, 1.5], size=(T, n, 3)).astype(np.float32)
a = torch.from_numpy(a).cuda()
for t in range(T):
    env.step(a[t])
torch.cuda.synchronize()
print(mode, env.counters())
