"""Dev tool: per-step time of the fused rollout kernel vs single-step launches."""
import os, sys, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
from reinforcementlearning4meshgeneration_amd.domains import boundary
n=4096
env=MeshVecEnv([boundary(0)], n_envs=n)
g=torch.Generator(device='cuda'); g.manual_seed(0)
lo=torch.tensor([-1.,-1.5,0.],device='cuda'); hi=torch.tensor([1.,1.5,1.5],device='cuda')
for T in (1,2,4,16,64,256):
    a=(lo+(hi-lo)*torch.rand((T,n,3),device='cuda',generator=g)).float().contiguous()
    env.rollout(a); torch.cuda.synchronize()
    env.set_timing(1)
    for r in range(5): env.rollout(a)
    kt=env.kernel_times_ms(); env.set_timing(0)
    print(f"rollout T={T:4d}: kernel {kt.mean()*1e3:9.1f} us  -> {kt.mean()*1e3/T:7.2f} us/step  -> {n*T/(kt.mean()*1e-3):.3e} env-steps/s")
