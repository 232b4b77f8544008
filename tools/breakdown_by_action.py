"""Dev tool: kernel time by action mix (GPU)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
from reinforcementlearning4meshgeneration_amd.domains import boundary, read_polygon
n=4096; T=120
def run(name, doms, a0_mode):
    env=MeshVecEnv(doms, n_envs=n)
    g=torch.Generator(device='cuda'); g.manual_seed(0)
    lo=torch.tensor([-1.,-1.5,0.],device='cuda'); hi=torch.tensor([1.,1.5,1.5],device='cuda')
    a=(lo+(hi-lo)*torch.rand((T,n,3),device='cuda',generator=g)).float()
    if a0_mode is not None: a[:,:,0]=a0_mode
    a=a.contiguous()
    for t in range(20): env.step(a[t])
    torch.cuda.synchronize(); c0=env.counters(); env.set_timing(1)
    for t in range(20,T): env.step(a[t])
    kt=env.kernel_times_ms(); c1=env.counters()
    st=c1['steps']-c0['steps']; v=c1['valid']-c0['valid']
    print(f"{name:28s} kern_us avg={kt.mean()*1e3:7.2f} min={kt.min()*1e3:7.2f} valid_rate={v/st:.3f} mean_n={(c1['sum_ring']-c0['sum_ring'])/st:.1f}")
    env.close()
b0=[boundary(0)]
run("trivial (n0=5 done path)", [[(0,0),(0,1),(1,1),(1.5,0.5),(1,0)]], None)
run("boundary0 uniform", b0, None)
run("boundary0 rule -1 only", b0, -1.0)
run("boundary0 rule +1 only", b0, 1.0)
run("boundary0 rule 0 only", b0, 0.0)
