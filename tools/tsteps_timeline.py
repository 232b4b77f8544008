"""Dev tool: where a workgroup of the T-step closed-loop kernel (k_step_group_actor_T) spends a step.
Stamps per workgroup and step: 0 step begins | 1 this workgroup's wave 0 reached the barrier after the env step |
2 barrier released (slowest wave done) | 3 actor forward + closing barrier done."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinforcementlearning4meshgeneration_amd import MeshVecEnv, boundary
from reinforcementlearning4meshgeneration_amd.actor import FusedActor
n, T = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 32
dom = boundary(0)
if len(sys.argv) > 2 and sys.argv[2] == "d1":
    dom = [tuple(p) for p in np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "boundary16_biased_s2.npz"))["domain_xy"]]
torch.manual_seed(999)
lin = [torch.nn.Linear(18, 128), torch.nn.Linear(128, 128), torch.nn.Linear(128, 128)]
mu, ls = torch.nn.Linear(128, 3), torch.nn.Linear(128, 3)
actor = FusedActor.from_torch(lin, mu, ls)
env = MeshVecEnv([dom], n_envs=n)
env.reset()
nxt = actor.sample(env.obs, seed=1, counter=0).clone()
for k in range(6):
    h = env.step_actor_T(actor, nxt, T, seed=1, counter=1 + k * T); nxt = h["actions"][T].clone()
dbg = torch.zeros((n // 16, T, 4), dtype=torch.int64, device="cuda")
import ctypes
from reinforcementlearning4meshgeneration_amd import _capi
L = _capi.load()     # MESHENV_LIB must point at a -DMESHENV_DEV build (tools/build_variant.sh dev -DMESHENV_DEV)
L.meshenv_dev_set_tsteps_dbg.argtypes = [ctypes.c_void_p]
L.meshenv_dev_set_tsteps_dbg(dbg.data_ptr())
h = env.step_actor_T(actor, nxt, T, seed=1, counter=1000)
torch.cuda.synchronize()
L.meshenv_dev_set_tsteps_dbg(None)
d = dbg.cpu().numpy().astype(np.float64) * 0.01   # us
valid_per_wg = None
body = d[:, :, 1] - d[:, :, 0]; wait = d[:, :, 2] - d[:, :, 1]; act = d[:, :, 3] - d[:, :, 2]; step = d[:, :, 3] - d[:, :, 0]
gap = d[:, 1:, 0] - d[:, :-1, 3]
print(f"T={T}: per workgroup-step (us): wave0 body {body.mean():.2f}  wait for slowest wave {wait.mean():.2f}  actor + barrier {act.mean():.2f}  "
      f"whole step {step.mean():.2f} (p10 {np.percentile(step,10):.2f} p50 {np.percentile(step,50):.2f} p90 {np.percentile(step,90):.2f})  gap between steps {gap.mean():.3f}")
tot = d[:, -1, 3] - d[:, 0, 0]
print(f"per workgroup total {tot.mean():.1f} us (min {tot.min():.1f} max {tot.max():.1f}); launch span {(d[:, -1, 3].max() - d[:, 0, 0].min()):.1f} us = {(d[:, -1, 3].max() - d[:, 0, 0].min()) / T:.2f} us per step")
