"""Dev tool: phase 1 of the CU-group kernel by outcome class (diagnostic build, build_variants/dbg_stamps.so =
tools/build_variant.sh with -DMESHENV_STAMPS).  For each wave: entry (t0) relative to the launch's first wave, time in its
own check + store (t1 - t0), and for each workgroup when its barrier is released; classes from the step's results.
usage: python tools/phase1_classes.py"""
import os, sys, ctypes as C, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MESHENV_LIB"] = os.path.join(ROOT, "build_variants/dbg_stamps.so")
os.environ["MESHENV_GROUP"] = "16"
sys.path.insert(0, ROOT)
from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
from reinforcementlearning4meshgeneration_amd.domains import boundary
n = 4096
env = MeshVecEnv([boundary(0)], n_envs=n)
env.reset()
g = torch.Generator(device='cuda'); g.manual_seed(0)
lo = torch.tensor([-1., -1.5, 0.], device='cuda'); hi = torch.tensor([1., 1.5, 1.5], device='cuda')
T = 120
a = (lo + (hi - lo) * torch.rand((T, n, 3), device='cuda', generator=g)).float().contiguous()
env._L.meshenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
tick = 0.01
acc = {}
skew, rel_all, first_all = [], [], []
for t in range(T):
    c0 = env.counters()
    o, r, d, cpl = env.step(a[t])
    if t < 40: continue
    torch.cuda.synchronize()
    st = np.zeros(n * 16, np.uint64)
    env._L.meshenv_debug_stamps(env._handle, st.ctypes.data)
    st = st.reshape(n, 16).astype(np.int64)
    t0, t1, t2, pend = st[:, 0], st[:, 1], st[:, 2], st[:, 3]
    base = t0.min()
    act = a[t].cpu().numpy()
    rule = np.where(act[:, 0] <= -0.5, -1, np.where(act[:, 0] >= 0.5, 1, 0))
    cls = np.where(pend == 1, np.where(rule == 0, "valid rule 0", "valid rule +-1"), np.where(rule == 0, "rejected rule 0", "rejected rule +-1"))
    for k in np.unique(cls):
        m = cls == k
        acc.setdefault(k, []).append(((t1 - t0)[m] * tick))
    wg0 = t0.reshape(-1, 16).min(axis=1)
    skew.append((wg0 - base) * tick)
    rel_all.append((t2.reshape(-1, 16).max(axis=1) - wg0) * tick)
    first_all.append(((t0.reshape(-1, 16).max(axis=1)) - wg0) * tick)
skew = np.concatenate(skew); rel = np.concatenate(rel_all); fa = np.concatenate(first_all)
print("workgroup entry after the launch's first wave: p50 %.2f p90 %.2f max %.2f us" % (np.percentile(skew, 50), np.percentile(skew, 90), skew.max()))
print("last wave of a workgroup enters after its first: p50 %.2f p90 %.2f us" % (np.percentile(fa, 50), np.percentile(fa, 90)))
print("barrier release after the workgroup's own first wave: p10 %.2f p50 %.2f p90 %.2f max %.2f us" % tuple(np.percentile(rel, [10, 50, 90, 100])))
for k, v in sorted(acc.items()):
    v = np.concatenate(v)
    print("%-18s %6.1f %% of waves, own phase 1 (entry -> at the barrier): p10 %.2f p50 %.2f p90 %.2f p99 %.2f us" % (k, 100 * len(v) / ((T - 40) * n), *np.percentile(v, [10, 50, 90, 99])))
