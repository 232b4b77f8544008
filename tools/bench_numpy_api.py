"""Dev tool: rate of the SB3-shaped numpy API (step_async / step_wait: actions from host memory, results back to host,
infos built) -- the PCIe-inclusive figure DESIGN.md quotes next to the device-resident headline."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reinforcementlearning4meshgeneration_amd import MeshVecEnv, boundary
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for lazy in (True, False):
    env = MeshVecEnv([boundary(0)], n_envs=n, lazy_infos=lazy)
    env.reset_numpy()
    rng = np.random.default_rng(0)
    acts = rng.uniform([-1, -1.5, 0], [1, 1.5, 1.5], size=(300, n, 3)).astype(np.float32)
    for t in range(50):
        env.step_async(acts[t]); env.step_wait()
    t0 = time.perf_counter()
    for t in range(50, 300):
        env.step_async(acts[t]); obs, rew, done, infos = env.step_wait()
    dt = time.perf_counter() - t0
    print(f"lazy_infos={lazy}: {250 * n / dt:.3e} env-steps/s, {1e6 * dt / 250:.1f} us per step_wait ({n} envs)")
    env.close()
