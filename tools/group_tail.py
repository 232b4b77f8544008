"""Dev tool: what makes the slowest workgroups of the CU-group kernel slow (diagnostic build)."""
import os, sys, ctypes as C, numpy as np, torch
os.environ["MESHENV_LIB"]=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),"build_variants/dbg_stamps.so")
os.environ["MESHENV_GROUP"]="16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
from reinforcementlearning4meshgeneration_amd.domains import boundary
n=4096
env=MeshVecEnv([boundary(0)], n_envs=n)
g=torch.Generator(device='cuda'); g.manual_seed(0)
lo=torch.tensor([-1.,-1.5,0.],device='cuda'); hi=torch.tensor([1.,1.5,1.5],device='cuda')
T=90
a=(lo+(hi-lo)*torch.rand((T,n,3),device='cuda',generator=g)).float().contiguous()
env._L.meshenv_debug_stamps.argtypes=[C.c_void_p,C.c_void_p]
rows=[]; acc={}
for t in range(T):
    env.step(a[t])
    if t < 30: continue
    torch.cuda.synchronize()
    st=np.zeros(n*16,np.uint64)
    env._L.meshenv_debug_stamps(env._handle, st.ctypes.data)
    st=st.reshape(n,16).astype(np.int64); tick=0.01
    base=st[:,0].min()
    pend=st[:,3]==1
    upd=st[:,15]>0
    wg=np.arange(n)//16
    npend=np.bincount(wg[pend], minlength=n//16)
    end=np.zeros(n//16); rel=np.zeros(n//16); dur=np.zeros(n//16)
    for w in range(n//16):
        sl=slice(16*w,16*w+16)
        rel[w]=(st[sl,2].max()-base)*tick
        u=upd[sl]
        end[w]=((st[sl,15][u].max()-base)*tick) if u.any() else (st[sl,2].max()-base)*tick
        dur[w]=((st[sl,15][u]-st[sl,7][u]).max()*tick) if u.any() else 0
    k=int(np.argmax(end))
    rows.append((end.max(), np.percentile(end,50), np.percentile(end,90), np.percentile(end,99), npend[k], rel[k], dur[k], npend.max()))
    for w in range(n//16):
        acc.setdefault(int(npend[w]), []).append((rel[w], dur[w], end[w]))
r=np.array(rows)
for k in sorted(acc):
    v=np.array(acc[k]); print("pending %d: WGs %5d  barrier release %.2f  longest update %.2f  end %.2f"%(k, len(v), v[:,0].mean(), v[:,1].mean(), v[:,2].mean()))
print("per launch: kernel end max %.2f (mean over launches) | p50 %.2f p90 %.2f p99 %.2f of per-WG end times"%(r[:,0].mean(), r[:,1].mean(), r[:,2].mean(), r[:,3].mean()))
print("slowest WG: pending %.1f (max pending anywhere %.1f), barrier release %.2f, longest update %.2f"%(r[:,4].mean(), r[:,7].mean(), r[:,5].mean(), r[:,6].mean()))
