"""Dev tool: event counts and timeline of k_step_spec (diagnostic build: tools/build_variant.sh specstats -DMESHENV_SPEC_STATS)."""
import os, sys, ctypes as C, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MESHENV_LIB"] = os.path.join(ROOT, "build_variants/lib_specstats.so")
sys.path.insert(0, ROOT)
from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
from reinforcementlearning4meshgeneration_amd.domains import boundary
n = 4096
env = MeshVecEnv([boundary(0)], n_envs=n)
print(env.step_kernel)
g = torch.Generator(device='cuda'); g.manual_seed(0)
lo = torch.tensor([-1., -1.5, 0.], device='cuda'); hi = torch.tensor([1., 1.5, 1.5], device='cuda')
T = 300
a = (lo + (hi - lo) * torch.rand((T, n, 3), device='cuda', generator=g)).float().contiguous()
L = env._L
L.meshenv_debug_spec_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
for t in range(200): env.step(a[t])
cnt = np.zeros(16, np.uint64); tm = np.zeros(n * 12, np.uint64)
L.meshenv_debug_spec_stats(env._handle, cnt.ctypes.data, None, n, 1)
K = 50
for t in range(200, 200 + K): env.step(a[t])
L.meshenv_debug_spec_stats(env._handle, cnt.ctypes.data, None, n, 1)
if cnt.sum() == 0: print("(timestamp build: no counts; build with -DMESHENV_SPEC_STATS=2 for them)")
names = ["posted", "claimed", "discard@quad", "discard@end", "committed", "valid", "valid self-applied"]
print({k: float(cnt[i]) / K for i, k in enumerate(names)})
# timeline of the LAST of a run of back-to-back launches (a launch on an idle GPU starts its workgroups over ~4 us)
for t in range(250, 300): env.step(a[t])
L.meshenv_debug_spec_stats(env._handle, cnt.ctypes.data, tm.ctypes.data, n, 1)
tm = tm.reshape(n, 12).astype(np.int64); tick = 0.01
base = tm[:, 8].max() - 100          # entries of the last launch lie within a microsecond of each other
tm[tm < base] = 0                    # stamps left over from earlier launches
base = tm[:, 8][tm[:, 8] > 0].min()
def st(name, v):
    v = np.asarray(v, float)
    if v.size == 0: print(f"{name:34s}: none"); return
    print(f"{name:34s}: n {v.size:5d} mean {v.mean():6.2f} p50 {np.percentile(v,50):6.2f} p90 {np.percentile(v,90):6.2f} max {v.max():6.2f}")
R = lambda k, m: (tm[m, k] - base) * tick
allw = tm[:, 8] > 0
st("kernel entry", R(8, allw))
if "spec" in env.step_kernel:
    st("after the t=0 barrier", R(9, allw))
    st("ring staged (load done)", R(0, tm[:, 0] > 0))
    posted = tm[:, 1] > 0; claimed = posted & (tm[:, 2] > 0); upd = claimed & (tm[:, 3] > 0)
    comm = tm[:, 5] > 0; selfd = tm[:, 6] > 0; rewd = tm[:, 7] > 0
    st("post", R(1, posted))
    st("claim - post", (tm[claimed, 2] - tm[claimed, 1]) * tick)
    st("update done - claim", (tm[upd, 3] - tm[upd, 2]) * tick)
    st("owner verdict, posted envs", R(4, posted))
    st("owner verdict, all envs", R(4, tm[:, 4] > 0))
    st("commit: update wave stored", R(5, comm))
    st("commit: owner wrote the reward", R(7, rewd))
    st("no commit: owner finished", R(6, selfd))
    late = selfd & (R(6, np.ones(n, bool)) > 9)
    print("owners finishing after 9 us without commit (self-applied extractions):", int(late.sum()))

    # the envs that end the launch
    endt = np.where(comm, np.maximum(R(5, np.ones(n, bool)), R(7, np.ones(n, bool))), np.where(selfd, R(6, np.ones(n, bool)), 0))
    wg_posts = np.bincount(np.arange(n)[posted] // 16, minlength=n // 16)
    print("launch end (max over envs): %.2f ; per-WG posts: mean %.2f max %d" % (endt.max(), wg_posts.mean(), wg_posts.max()))
    print(" env   WGposts  loaded  post  claim  upd_done  verdict  stored  reward  self_end")
    for e in np.argsort(-endt)[:14]:
        f = lambda k: ("%6.2f" % ((tm[e, k] - base) * tick)) if tm[e, k] > 0 else "   -  "
        print("%5d  %4d    %s %s %s %s %s %s %s %s" % (e, wg_posts[e // 16], f(0), f(1), f(2), f(3), f(4), f(5), f(7), f(6)))
    for k in range(0, 9):
        sel = wg_posts[np.arange(n) // 16] == k
        v = endt[sel & (endt > 0)]
        if v.size: print("WGs with %d posts: envs %5d  end mean %.2f max %.2f" % (k, v.size, v.mean(), v.max()))
