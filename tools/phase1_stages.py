"""Dev tool: the check phase of the CU-group kernel stage by stage, for the waves whose action is valid (diagnostic builds
build_variants/dbg_p1_0.so / dbg_p1_1.so = tools/build_variant.sh with -DMESHENV_STAMPS -DMESHENV_DBG_P1SET=0 / 1).
usage: python tools/phase1_stages.py"""
import os, sys, ctypes as C, subprocess, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] in ("0", "1"):
    os.environ["MESHENV_LIB"] = os.path.join(ROOT, f"build_variants/dbg_p1_{sys.argv[1]}.so")
    os.environ["MESHENV_GROUP"] = "16"
    sys.path.insert(0, ROOT)
    import torch
    from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
    from reinforcementlearning4meshgeneration_amd.domains import boundary
    n = 4096
    dom = boundary(0)
    if len(sys.argv) > 2 and sys.argv[2] != "boundary0":
        name = {"d1": "boundary16_biased_s2", "d2": "boundary15_biased_s5", "d3": "test1_biased_s42"}.get(sys.argv[2], sys.argv[2])
        dom = [tuple(p) for p in np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))["domain_xy"]]
    env = MeshVecEnv([dom], n_envs=n); env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    lo = torch.tensor([-1., -1.5, 0.], device='cuda'); hi = torch.tensor([1., 1.5, 1.5], device='cuda')
    a = (lo + (hi - lo) * torch.rand((60, n, 3), device='cuda', generator=g)).float().contiguous()
    for t in range(60): env.step(a[t])
    torch.cuda.synchronize()
    st = np.zeros(n * 16, np.uint64)
    env._L.meshenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    env._L.meshenv_debug_stamps(env._handle, st.ctypes.data)
    st = st.reshape(n, 16).astype(np.int64)
    act = a[59].cpu().numpy()
    rule0 = (act[:, 0] > -0.5) & (act[:, 0] < 0.5)
    tick = 0.01
    da, db, de = (st[:, 4] - st[:, 0]) * tick, (st[:, 5] - st[:, 0]) * tick, (st[:, 1] - st[:, 0]) * tick
    m = (st[:, 3] == 1) & rule0 & (da > 0) & (da < 30) & (db > 0) & (db < 30)   # LDS scratch is not cleared: stale stamps fall outside
    print(json.dumps(dict(n=int(m.sum()), of=int(((st[:, 3] == 1) & rule0).sum()), a=float(np.median(da[m])), b=float(np.median(db[m])), end=float(np.median(de[m])))))
else:
    dom_arg = sys.argv[1:2]      # usage: python tools/phase1_stages.py [boundary0 | d1 | d2 | d3]
    r = [json.loads(subprocess.run([sys.executable, __file__, str(k)] + dom_arg, capture_output=True, text=True).stdout.strip().splitlines()[-1]) for k in (0, 1)]
    d, p, q, x, e = r[0]["a"], r[0]["b"], r[1]["a"], r[1]["b"], r[1]["end"]
    print(f"valid rule-0 waves ({r[0]['n']} of {r[0]['of']}, medians): entry -> decoded {d:.2f} us | ring pass {p - d:.2f} | quad stage {q - p:.2f} | intersections {x - q:.2f} | hand-over {e - x:.2f} | at the barrier {e:.2f}")
