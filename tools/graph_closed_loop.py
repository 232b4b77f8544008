"""Dev tool: does a HIP graph of K one-launch closed-loop steps (meshenv_step_actor) beat K stream launches?
(timing only: the noise counters are baked into the captured nodes)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinforcementlearning4meshgeneration_amd import MeshVecEnv
from reinforcementlearning4meshgeneration_amd.actor import FusedActor
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dom = [tuple(p) for p in np.load(os.path.join(root, "tests", "golden", "boundary16_biased_s2.npz"))["domain_xy"]]
n, K = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(999)
lin = [torch.nn.Linear(18, 128), torch.nn.Linear(128, 128), torch.nn.Linear(128, 128)]
mu, ls = torch.nn.Linear(128, 3), torch.nn.Linear(128, 3)
actor = FusedActor.from_torch(lin, mu, ls)
env = MeshVecEnv([dom], n_envs=n)
env.reset()
nxt = actor.sample(env.obs, seed=1, counter=0).clone()
def chunk(nxt, c0):
    for t in range(K):
        _, _, _, _, nxt = env.step_actor(actor, nxt, seed=1, counter=c0 + t)
    return nxt
for k in range(8):
    nxt = chunk(nxt, 1 + k * K)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(20):
    nxt = chunk(nxt, 1000 + k * K)
torch.cuda.synchronize()
print(f"stream launches: {1e6 * (time.perf_counter() - t0) / (20 * K):.2f} us per vector step")
import ctypes as C
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
pp = [torch.empty((n, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
pp[0].copy_(nxt)
with torch.cuda.stream(s):
    env._bind_stream()                      # binds env (and below the actor) to s before the capture starts
    actor._L.meshenv_actor_set_stream(actor._h, C.c_void_p(s.cuda_stream)); actor._stream = s.cuda_stream
    def raw_chunk(c0):
        for t in range(K):
            rc = env._L.meshenv_step_actor(env._handle, actor._h, pp[t & 1].data_ptr(), env.obs.data_ptr(), env.reward.data_ptr(),
                                           env.done.data_ptr(), env.complete.data_ptr(), env.terminal_obs.data_ptr(), 1, 1,
                                           C.c_uint64(1), C.c_uint64(c0 + t), pp[(t + 1) & 1].data_ptr(), None)
            assert rc == 0, rc
    raw_chunk(5000)
    s.synchronize()
    t0 = time.perf_counter()
    for k in range(20): raw_chunk(7000 + k * K)
    s.synchronize()
    print(f"raw C-ABI stream launches: {1e6 * (time.perf_counter() - t0) / (20 * K):.2f} us per vector step")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        raw_chunk(6000)
    s.synchronize()
    for _ in range(3): g.replay()
    s.synchronize()
    t0 = time.perf_counter()
    for k in range(20): g.replay()
    s.synchronize()
    print(f"graph of {K} nodes: {1e6 * (time.perf_counter() - t0) / (20 * K):.2f} us per vector step")
