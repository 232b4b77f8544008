"""Dev tool: per-wave timeline of one k_step launch (diagnostic build with -DMESHENV_STAMPS)."""
import os, sys, ctypes as C, numpy as np, torch
os.environ["MESHENV_LIB"]=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),"build_variants/dbg_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
from reinforcementlearning4meshgeneration_amd.domains import boundary
n=int(sys.argv[1]) if len(sys.argv)>1 else 4096
env=MeshVecEnv([boundary(0)], n_envs=n)
g=torch.Generator(device='cuda'); g.manual_seed(0)
lo=torch.tensor([-1.,-1.5,0.],device='cuda'); hi=torch.tensor([1.,1.5,1.5],device='cuda')
a=(lo+(hi-lo)*torch.rand((60,n,3),device='cuda',generator=g)).float().contiguous()
for t in range(60):
    env.step(a[t])
torch.cuda.synchronize()
raw=np.zeros(n*4,np.uint64)
env._L.meshenv_debug_raw_counters.argtypes=[C.c_void_p,C.c_void_p]
env._L.meshenv_debug_raw_counters(env._handle, raw.ctypes.data)
raw=raw.reshape(n,4).astype(np.int64)
t0=raw[:,0]; t1=raw[:,1]; t2=raw[:,2]; valid=(raw[:,3]&0xff)>0; rule=(raw[:,3]>>8)&3
base=t0.min()
tick=0.01 # us per tick (100 MHz)
print("kernel span us:", (t2.max()-base)*tick)
print("start times us: p50=%.2f p90=%.2f max=%.2f"%tuple(np.percentile((t0-base)*tick,[50,90,100])))
print("load phase us: mean=%.2f p90=%.2f max=%.2f"%(((t1-t0)*tick).mean(), np.percentile((t1-t0)*tick,90), ((t1-t0)*tick).max()))
dur=(t2-t0)*tick
for name,m in [("all",np.ones(n,bool)),("valid",valid),("fail rule0",(~valid)&(rule==0)),("fail rule-1",(~valid)&(rule==1)),("fail rule+1",(~valid)&(rule==2))]:
    if m.sum(): print(f"{name:12s} n={m.sum():5d} wave dur us: mean={dur[m].mean():6.2f} p50={np.percentile(dur[m],50):6.2f} p90={np.percentile(dur[m],90):6.2f} max={dur[m].max():6.2f}  end-time max={((t2[m]-base)*tick).max():6.2f}")
comp=(t2-t1)*tick
for name,m in [("valid",valid),("fail rule0",(~valid)&(rule==0)),("fail rule+-1",(~valid)&(rule!=0))]:
    if m.sum(): print(f"{name:12s} compute-phase us: mean={comp[m].mean():6.2f} p50={np.percentile(comp[m],50):6.2f} max={comp[m].max():6.2f}")
st=np.zeros(n*16,np.uint64)
env._L.meshenv_debug_stamps.argtypes=[C.c_void_p,C.c_void_p]
env._L.meshenv_debug_stamps(env._handle, st.ctypes.data)
st=st.reshape(n,16).astype(np.int64)
names={1:"decode",2:"PIP",3:"same-pt+quad setup",4:"quad_valid",5:"intersects",6:"area/robust+update",7:"V2 stage+keys",8:"bq",9:"select+prep",10:"stage A",11:"stage B",12:"stage C",13:"reductions",14:"stage D+final"}
for label,m in [("VALID rule0",valid&(rule==0)),("VALID rule+-1",valid&(rule!=0))]:
    if not m.sum(): continue
    print(label, "n=",m.sum())
    prev=t1[m]
    for k in range(1,15):
        cur=st[m,k]
        ok=cur>0
        if ok.sum()==0: continue
        d=(cur-prev)[ok]*tick
        print(f"   {k:2d} {names[k]:22s} mean={d.mean():6.2f} us  (n={ok.sum()})")
        prev=np.where(ok,cur,prev)
    print("   end tail mean=%.2f"%(((t2[m]-prev)*tick).mean()))
cyc=st[:,15].astype(np.float64); durr=(t2-t0).astype(np.float64)*10.0  # ns
print("shader clock GHz (s_memtime/s_memrealtime): mean=%.3f p10=%.3f p90=%.3f"%((cyc/durr).mean(), np.percentile(cyc/durr,10), np.percentile(cyc/durr,90)))
order=np.argsort(-dur)[:12]
print("slowest waves: env dur start valid rule stage-deltas")
for e in order:
    prev=t1[e]; ds=[]
    for k in range(1,15):
        if st[e,k]>0: ds.append(f"{k}:{(st[e,k]-prev)*tick:.1f}"); prev=st[e,k]
    print(e, f"{dur[e]:.1f}", f"{(t0[e]-base)*tick:.1f}", int(valid[e]), int(rule[e]), " ".join(ds), f"tail:{(t2[e]-prev)*tick:.1f}")
print("valid dur percentiles 50/90/99/100:", np.percentile(dur[valid],[50,90,99,100]))
hw=st[:,0]
# gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13](+) ...; use all bits above wave_id as the SIMD key
simd_key=((hw>>4)&0xfff) | ((hw>>32)<<12)   # simd, pipe, cu, sh, se + xcc
import collections
cntv=collections.Counter(simd_key[valid].tolist())
share=np.array([cntv[k] for k in simd_key])
for m in sorted(set(share[valid].tolist())):
    sel=valid&(share==m)
    print(f"valid waves with {m} valid wave(s) on their SIMD: n={sel.sum():4d} dur mean={dur[sel].mean():6.2f} max={dur[sel].max():6.2f}")
print("distinct SIMD keys:", len(set(simd_key.tolist())))
