"""Dev tool: phase-2 stage timeline of the CU-group kernel (diagnostic build)."""
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
v=st[:,15]>0
print("updated envs:", v.sum())
names={6:"update (to stamp 6)",9:"select",10:"stage A",11:"stage B",12:"stage C",13:"reductions",14:"rows/final"}
prev=st[v,7]
for k in (6,9,10,11,12,13,14):
    cur=st[v,k]; print(f"  {names[k]:22s} {((cur-prev)*tick).mean():5.2f} us"); prev=cur
print(f"  apply tail (reward)     {((st[v,8]-prev)*tick).mean():5.2f} us")
print(f"  finish_and_store        {((st[v,15]-st[v,8])*tick).mean():5.2f} us")
print(f"  total phase 2           {((st[v,15]-st[v,7])*tick).mean():5.2f} us")
