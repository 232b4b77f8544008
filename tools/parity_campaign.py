"""Dev tool: large GPU-vs-oracle parity campaign (millions of steps); prints mismatch statistics."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_lib import RefBatch, RefEnv
from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
from reinforcementlearning4meshgeneration_amd.domains import boundary, random_domain

def golden_domain(name):
    tr = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return [tuple(p) for p in tr["domain_xy"]]

def campaign(label, doms, env_domain, T, seed, biased, threads=16):
    n = len(env_domain)
    env = MeshVecEnv(doms, env_domain=env_domain)
    refs = [RefEnv(np.asarray(doms[d], np.float64), env.constants[d].original_area, env.constants[d].est_min_l,
                   env.constants[d].est_crit_l, cap_new=64) for d in env_domain]
    batch = RefBatch(refs); batch.reset(); env.reset()
    rng = np.random.default_rng(seed)
    st = dict(obs_mis=0, obs_tot=0, max_obs=0.0, max_rew=0.0, flag_mis=0, topo_checked=0, topo_bad=0, valid=0, done=0)
    t0 = time.time()
    for t in range(T):
        a = rng.uniform([-1, -1.5, 0], [1, 1.5, 1.5], size=(n, 3))
        if biased:
            pick = rng.random(n) < 0.6
            b = np.stack([rng.uniform(-1, 1, n), rng.uniform(0.2, 1.0, n), rng.uniform(0.3, 1.2, n)], axis=1)
            a[pick] = b[pick]
        a = a.astype(np.float32)
        o, r, d, c = env.step(torch.from_numpy(a).cuda())
        o = o.cpu().numpy(); r = r.cpu().numpy(); d = d.cpu().numpy(); c = c.cpu().numpy()
        o_ref, r_ref, d_ref, c_ref = batch.step(a, auto_reset=True, threads=threads)
        st["obs_mis"] += int((o != o_ref).sum()); st["obs_tot"] += o.size
        st["max_obs"] = max(st["max_obs"], float(np.abs(o.astype(np.float64) - o_ref).max()))
        st["max_rew"] = max(st["max_rew"], float(np.abs(r - r_ref).max()))
        st["flag_mis"] += int((d != d_ref).sum() + (c != c_ref).sum()); st["done"] += int(d_ref.sum())
        if t % 100 == 99 or t == T - 1:
            for k in rng.choice(n, size=min(128, n), replace=False):
                s = env.get_state(int(k)); ids, xy = refs[k].ring(); cid, _ = refs[k].candidates()
                st["topo_checked"] += 1
                if not (np.array_equal(s["ring_ids"], ids) and np.array_equal(s["ring_xy"], xy) and np.array_equal(s["cand_order_ids"], cid)
                        and s["ref_id"] == refs[k].ref_id()):
                    st["topo_bad"] += 1
    st["valid"] = env.counters()["valid"]; st["steps"] = n * T; st["seconds"] = round(time.time() - t0, 1)
    print(label, st, flush=True)
    env.close()

def move_campaign(label, doms, env_domain, T, seed):
    """move() API (rl/boundary_env.py:265-432) in lockstep with the oracle's meshenv_ref_move, including the moves that
    go through smooth_pave (:405-426)."""
    n = len(env_domain)
    env = MeshVecEnv(doms, env_domain=env_domain, auto_reset=False, log_capacity=512)
    refs = [RefEnv.from_points(doms[d], cap_new=512) for d in env_domain]
    obs = env.reset(static=True).cpu().numpy()
    assert np.array_equal(obs, np.stack([r.reset(static=True)[0] for r in refs]))
    rng = np.random.default_rng(seed)
    st = dict(moves=0, codes=[0, 0, 0, 0, 0], smoothed=0, valid=0, obs_mis=0, obs_mis_moves=0, obs_mis_domains=set(), max_obs=0.0, flag_mis=0, topo_checked=0, topo_bad=0,
              max_ring_dev=0.0)
    t0 = time.time()
    for t in range(T):
        pts = np.stack([rng.uniform(0.05, 0.45, n), rng.uniform(0.2, 1.5, n)], axis=1)
        typ = rng.uniform(0, 1, n)
        o, d, c, code = [x.cpu().numpy() for x in env.move(torch.from_numpy(pts), torch.from_numpy(typ))]
        mask = np.zeros(n, np.uint8)
        for k in range(n):
            ne0 = refs[k].scalars()["n_elem"]
            nv0 = refs[k].not_valid_count()
            o_r, d_r, c_r, code_r = refs[k].move(pts[k], typ[k])
            st["smoothed"] += int(code_r != 2 and refs[k].scalars()["n_elem"] == ne0 and refs[k].not_valid_count() == 0 and nv0 > 0)
            st["codes"][code_r] += 1
            bad = int(code[k] != code_r) + (int(bool(d[k]) != d_r) + int(bool(c[k]) != c_r) if code_r != 2 else 0)
            st["flag_mis"] += bad
            if bad:
                print(f"  flag mismatch: step {t} env {k} domain {int(env_domain[k])} code {int(code[k])}/{code_r} done {int(d[k])}/{int(d_r)} "
                      f"complete {int(c[k])}/{int(c_r)} front {len(refs[k].ring()[0])} not_valid before {nv0}", flush=True)
            if code_r == 0:
                st["obs_mis"] += int((o[k] != o_r).sum())
                if (o[k] != o_r).any():
                    st["obs_mis_moves"] += 1
                    st["obs_mis_domains"].add(int(env_domain[k]))
                st["max_obs"] = max(st["max_obs"], float(np.abs(o[k].astype(np.float64) - o_r).max()))
            st["valid"] += refs[k].scalars()["n_elem"] > ne0
            if d_r or code_r >= 2:
                mask[k] = 1
                refs[k].reset(static=True)
        st["moves"] += n
        if mask.any():
            env.reset(mask=torch.from_numpy(mask), static=True)
        if t % 50 == 49 or t == T - 1:
            for k in rng.choice(n, size=min(96, n), replace=False):
                s = env.get_state(int(k)); ids, xy = refs[k].ring()
                st["topo_checked"] += 1
                dev = float(np.abs(s["ring_xy"] - xy).max()) if len(ids) == len(s["ring_ids"]) else 1.0
                st["max_ring_dev"] = max(st["max_ring_dev"], dev)   # exact until a smoothing moved the front (tan / cos / sqrt)
                if not (np.array_equal(s["ring_ids"], ids) and dev <= 1e-10
                        and len(env.get_not_valid(int(k))) == refs[k].not_valid_count()):
                    st["topo_bad"] += 1
    st["seconds"] = round(time.time() - t0, 1)
    st["obs_mis_domains"] = sorted(st["obs_mis_domains"])
    print(label, st, flush=True)
    env.close()


if __name__ == "__main__":
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    seed_off = int(sys.argv[sys.argv.index("--seed-offset") + 1]) if "--seed-offset" in sys.argv else 0
    if "--move-only" in sys.argv:
        dom_off = int(sys.argv[sys.argv.index("--domain-offset") + 1]) if "--domain-offset" in sys.argv else 0   # other generated polygons
        mdoms = [boundary(0), boundary(-1), boundary(1), boundary(2)] + [golden_domain(x) for x in ("boundary16_biased_s2", "random1_1_biased_s1", "star_biased_s6")] + [random_domain(7000 + dom_off + k) for k in range(25)]
        move_campaign("move() x2048 on 32 domains", mdoms, (np.arange(2048) % len(mdoms)).astype(np.int32), int(60 * scale), 105 + seed_off)
        sys.exit(0)
    if "--long-only" in sys.argv:   # rings of several 64-vertex chunks only (the pre-filtered observation scan), other seeds
        big = [golden_domain(x) for x in ("boundary16_biased_s2", "boundary15_biased_s5", "test1_biased_s42", "dolphine3_biased_s0", "random1_1_biased_s1")]
        campaign("5 long shipped domains x2045 biased", big, (np.arange(2045) % 5).astype(np.int32), int(300 * scale), 203 + seed_off, True)
        campaign("5 long shipped domains x4096 biased (ragged CU-group kernel)", big, (np.arange(4096) % 5).astype(np.int32), int(150 * scale), 206 + seed_off, True)
        campaign("d1 x4096 uniform (CU-group kernel)", big[:1], np.zeros(4096, np.int32), int(150 * scale), 207 + seed_off, False)
        campaign("d3 x32768 biased (one wave per env)", big[2:3], np.zeros(32768, np.int32), int(20 * scale), 208 + seed_off, True)
        sys.exit(0)
    campaign("boundary0 x4096 uniform", [boundary(0)], np.zeros(4096, np.int32), int(1000 * scale), 101, False)
    campaign("boundary0 x4096 biased", [boundary(0)], np.zeros(4096, np.int32), int(1000 * scale), 102, True)
    big = [golden_domain(x) for x in ("boundary16_biased_s2", "boundary15_biased_s5", "test1_biased_s42", "dolphine3_biased_s0", "random1_1_biased_s1", "star_biased_s6")]
    campaign("6 shipped domains x2046 biased", big, (np.arange(2046) % 6).astype(np.int32), int(300 * scale), 103, True)
    # the same six rings (120 .. 272 vertices), sixteen per workgroup: the CU-group kernel with its LDS packed by ring length
    campaign("6 shipped domains x4096 biased (ragged CU-group kernel)", big, (np.arange(4096) % 6).astype(np.int32), int(150 * scale), 106, True)
    rnd = [random_domain(5000 + k) for k in range(1024)]
    campaign("1024 random polygons x4096 biased", rnd, (np.arange(4096) % 1024).astype(np.int32), int(400 * scale), 104, True)
    mdoms = [boundary(0), boundary(-1), boundary(1), boundary(2)] + big + [random_domain(7000 + k) for k in range(22)]
    if "--no-move" not in sys.argv:
        move_campaign("move() x2048 on 32 domains", mdoms, (np.arange(2048) % len(mdoms)).astype(np.int32), int(60 * scale), 105)
