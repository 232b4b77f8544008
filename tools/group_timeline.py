"""Dev tool: timeline of the CU-group kernel (diagnostic build)."""
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
a=(lo+(hi-lo)*torch.rand((60,n,3),device='cuda',generator=g)).float().contiguous()
for t in range(60): env.step(a[t])
torch.cuda.synchronize()
st=np.zeros(n*16,np.uint64)
env._L.meshenv_debug_stamps.argtypes=[C.c_void_p,C.c_void_p]
env._L.meshenv_debug_stamps(env._handle, st.ctypes.data)
st=st.reshape(n,16).astype(np.int64); tick=0.01
t0,t1,t2,pend,t4,who=st[:,0],st[:,1],st[:,2],st[:,3],st[:,4],st[:,5]
base=t0.min()
print("wave start: p50 %.2f max %.2f"%(np.percentile((t0-base)*tick,50), ((t0-base)*tick).max()))
print("arrive at barrier (t1-base): fail mean %.2f p90 %.2f max %.2f | pending mean %.2f p90 %.2f max %.2f"%(
  ((t1-base)*tick)[pend==0].mean(), np.percentile(((t1-base)*tick)[pend==0],90), ((t1-base)*tick)[pend==0].max(),
  ((t1-base)*tick)[pend==1].mean(), np.percentile(((t1-base)*tick)[pend==1],90), ((t1-base)*tick)[pend==1].max()))
print("barrier release (t2-base): mean %.2f p90 %.2f max %.2f"%(((t2-base)*tick).mean(), np.percentile((t2-base)*tick,90), ((t2-base)*tick).max()))
upd=who>0
print("updates run: %d ; update duration mean %.2f p90 %.2f max %.2f ; end-time max %.2f"%(upd.sum(), ((t4-t2)*tick)[upd].mean(), np.percentile(((t4-t2)*tick)[upd],90), ((t4-t2)*tick)[upd].max(), ((t4-base)*tick)[upd].max()))
