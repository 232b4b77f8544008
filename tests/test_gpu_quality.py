"""GPU: the finished-episode archive and the device-side quality report against the oracle.

Element lists / vertex tables: bit-exact.  Quality records: 1e-12 relative (the device squares with x*x where Python's
`** 2` is libm pow, and uses ocml transcendentals; everything else is the same operation order).  Statistics: 1e-9
(wave reduction order vs a sequential loop)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def _biased(rng, T, n):
    a = rng.uniform([-1, -1.5, 0], [1, 1.5, 1.5], size=(T, n, 3))
    pick = rng.random((T, n)) < 0.6
    b = np.stack([rng.uniform(-1, 1, (T, n)), rng.uniform(0.2, 1.0, (T, n)), rng.uniform(0.3, 1.2, (T, n))], axis=2)
    a[pick] = b[pick]
    return a.astype(np.float32)


def test_archive_and_quality_report_match_oracle():
    import torch
    from oracle.ref_lib import RefEnv, element_quality, quality_stats
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv, boundary, random_domain
    doms = [boundary(0), random_domain(7), random_domain(8)]
    n, T = 96, 420
    env_domain = (np.arange(n) % len(doms)).astype(np.int32)
    env = MeshVecEnv(doms, env_domain=env_domain, auto_reset=True, log_capacity=160)
    refs = [RefEnv(np.asarray(doms[d], np.float64), env.constants[d].original_area, env.constants[d].est_min_l,
                   env.constants[d].est_crit_l, cap_new=256) for d in env_domain]
    for r in refs:
        r.reset()
    env.reset()
    acts = _biased(np.random.default_rng(21), T, n)
    last = [None] * n          # (quads, vxy, complete) of the oracle's last finished episode with elements
    episodes = np.zeros(n, int)
    for t in range(T):
        _, _, d, c = env.step(torch.from_numpy(acts[t]).cuda())
        d = d.cpu().numpy(); c = c.cpu().numpy()
        for k in range(n):
            _, _, done, comp, _ = refs[k].step(acts[t, k])
            assert bool(done) == bool(d[k]) and bool(comp) == bool(c[k])
            if done:
                q, v = refs[k].elements()
                if len(q):
                    last[k] = (q.copy(), v.copy(), bool(comp))
                    episodes[k] += 1
                refs[k].reset()
    assert sum(e > 0 for e in episodes) > n // 2 and any(x is not None and x[2] for x in last)
    rec_l, st_l, cnt_l = [x.cpu().numpy() for x in env.element_quality("last")]
    rec_c, st_c, cnt_c = [x.cpu().numpy() for x in env.element_quality("current")]
    worst = 0.0
    for k in range(n):
        le = env.get_last_episode(k)
        assert le["episodes"] == episodes[k]
        if last[k] is None:
            assert len(le["quads"]) == 0 and cnt_l[k] == 0
        else:
            q, v, comp = last[k]
            np.testing.assert_array_equal(le["quads"], q)
            np.testing.assert_array_equal(le["vertex_xy"], v)
            assert le["is_complete"] == comp and not le["overflow"]
            assert cnt_l[k] == len(q)
            exp = element_quality(v[q])
            np.testing.assert_allclose(rec_l[k, :len(q)], exp, rtol=1e-12, atol=1e-13)
            worst = max(worst, float(np.abs(rec_l[k, :len(q)] - exp).max()))
            np.testing.assert_allclose(st_l[k], quality_stats(exp), rtol=1e-9, atol=1e-11)
            assert not rec_l[k, len(q):].any()
        # the running episode
        q, v = refs[k].elements()
        qd, vd = env.get_elements(k)
        np.testing.assert_array_equal(qd, q)
        np.testing.assert_array_equal(vd, v)
        assert cnt_c[k] == len(q)
        if len(q):
            exp = element_quality(v[q])
            np.testing.assert_allclose(rec_c[k, :len(q)], exp, rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(st_c[k], quality_stats(exp), rtol=1e-9, atol=1e-11)
    rep = env.quality_report("last")
    assert rep["meshes"] == int((cnt_l > 0).sum()) and rep["elements"] == int(cnt_l.sum())
    assert 0 < rep["stretch"]["average"] <= 1 and rep["min_angle_deg"]["min"] > 0
    print("quality: max |device - oracle| =", worst, {k: round(v["average"], 4) for k, v in rep.items() if isinstance(v, dict)})
    env.close()


def test_archive_survives_explicit_reset_and_fixture_json():
    """auto_reset = False: the caller resets after done (the reference's eval flow); the archive then holds the
    finished mesh, and its write_2_file JSON equals the reference's own file (fixture)."""
    import torch
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv, boundary
    from reinforcementlearning4meshgeneration_amd.export import mesh_graph
    fx = json.load(open(os.path.join(GOLDEN_DIR, "write2file_boundary0_biased_s1.json")))
    tr = np.load(os.path.join(GOLDEN_DIR, fx["trace"] + ".npz"))
    env = MeshVecEnv([boundary(0)], n_envs=1, auto_reset=False, log_capacity=128)
    env.reset()
    acts = torch.from_numpy(tr["actions"]).cuda()
    for t in range(fx["step"] + 1):
        _, _, d, c = env.step(acts[t:t + 1])
        if bool(d.cpu()[0]):
            if t == fx["step"]:
                assert bool(c.cpu()[0])
                cur_q, cur_v = env.get_elements(0)
            env.reset()
    le = env.get_last_episode(0)
    np.testing.assert_array_equal(le["quads"], cur_q)
    np.testing.assert_array_equal(le["vertex_xy"], cur_v)
    assert le["is_complete"]
    assert json.loads(json.dumps(mesh_graph(le["quads"], le["vertex_xy"], boundary(0)))) == fx["json"]
    q, _ = env.get_elements(0)
    assert len(q) == 0          # the new episode starts with an empty log
    env.close()


def test_full_size_invariants_4096_envs():
    """BASELINE.json configs[1] size (4096 x boundary()) through size-independent properties that tie the step kernel,
    the state accessors and the quality kernel together: area bookkeeping, ring-length bookkeeping, orientation, id
    uniqueness, validity limits of every accepted element."""
    import torch
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv, boundary
    n, T, n0 = 4096, 300, 30
    env = MeshVecEnv([boundary(0)], n_envs=n, auto_reset=True, log_capacity=192)
    env.reset()
    orig = env.constants[0].original_area
    acts = torch.from_numpy(_biased(np.random.default_rng(31), T, n)).cuda()
    for t in range(T):
        env.step(acts[t])
    rec, stats, cnt = [x.cpu().numpy() for x in env.element_quality("current")]
    checked = 0
    for k in range(0, n, 7):
        st = env.get_state(k)
        ne = int(cnt[k])
        assert ne == st["n_elem"]
        if ne == 0 or st["n"] <= 5:
            continue
        checked += 1
        area = orig
        for a in rec[k, :ne, 6]:
            area -= a                                   # current_area -= mesh_area, rl/boundary_env.py:200
        assert abs(area - st["current_area"]) <= 1e-11, k
        n_new = st["n_vert"] - n0
        assert st["n"] == n0 - 2 * (ne - n_new), k       # rule 0 keeps the ring length, rules -1/+1 remove two
        xy = st["ring_xy"]
        shoelace = 0.5 * float(np.sum(xy[:, 0] * np.roll(xy[:, 1], -1) - np.roll(xy[:, 0], -1) * xy[:, 1]))
        assert shoelace < 0 and abs(-shoelace - st["current_area"]) <= 2e-3 * orig, k   # clockwise; angles are 1e-4-rounded
        ids = st["ring_ids"]
        assert len(np.unique(ids)) == len(ids) and ids.min() >= 0 and ids.max() < st["n_vert"], k
        assert rec[k, :ne, 0].min() >= 1.8 - 1e-6 and rec[k, :ne, 1].max() <= 178.2 + 1e-6, k   # 0.01 pi .. 0.99 pi
        np.testing.assert_allclose(stats[k, 6, 1] * ne, rec[k, :ne, 6].sum(), rtol=1e-12)
    assert checked > 300
    le = env.quality_report("last")
    assert le["meshes"] > n // 2
    env.close()
