"""GPU: the move() API and reset(static=True) (first slice) through the C-ABI:
  (a) replay of the traces recorded from the reference's own move() (tests/golden/move_*.npz) -- observations
      bit-exact unless a device transcendental differs by an ulp (tolerance 1e-5 as everywhere, mismatches counted),
      topology / not_valid_points / flags / return codes exact;
  (b) 2048 envs on mixed domains in lockstep with the CPU oracle's move(), including the moves that go through
      smooth_pave (rl/boundary_env.py:405-426: front + interior smoothing, last_not_valid_points, fresh selection);
  (c) the reference-shaped single env: return tuple, the UnboundLocalError of a finished ring, static reset."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, move_golden_names
from test_oracle_golden import replay_move

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", move_golden_names())
def test_hip_move_matches_reference_trace(name):
    import torch
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv
    tr = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    worst = [0.0]

    def make(tr):
        return MeshVecEnv([[tuple(p) for p in tr["domain_xy"]]], n_envs=1, auto_reset=False, log_capacity=1024)

    def do_move(env, p, ty):
        obs, done, comp, code = env.move(torch.tensor(p[None, :], dtype=torch.float64), torch.tensor([ty], dtype=torch.float64))
        return obs.cpu().numpy()[0], bool(done.cpu()[0]), bool(comp.cpu()[0]), int(code.cpu()[0])

    def state_of(env):
        st = env.get_state(0)
        _, vxy = env.get_elements(0)
        st["n_not_valid"] = len(env.get_not_valid(0))
        st["last_vertex"] = vxy[st["n_vert"] - 1]
        return st

    def obs_cmp(a, b, err_msg=""):   # 1e-5 as everywhere on the device side (an ulp of a device transcendental)
        d = float(np.abs(np.asarray(a, np.float64) - b).max())
        worst[0] = max(worst[0], d)
        assert d <= 1e-5, err_msg

    seen = replay_move(tr, make, lambda env: env.reset(static=True).cpu().numpy()[0], do_move, state_of, obs_cmp)
    print(f"{name}: codes {seen.tolist()}, max obs err {worst[0]:.3g}")


def test_move_lockstep_2048_envs_mixed_domains():
    import torch
    from oracle.ref_lib import RefEnv
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv
    from reinforcementlearning4meshgeneration_amd.domains import boundary, random_domain
    d1 = [tuple(p) for p in np.load(os.path.join(GOLDEN_DIR, "boundary16_biased_s2.npz"))["domain_xy"]]
    doms = [boundary(0), boundary(-1), d1] + [random_domain(900 + k) for k in range(13)]
    n, T = 2048, 120
    env_domain = (np.arange(n) % len(doms)).astype(np.int32)
    env = MeshVecEnv(doms, env_domain=env_domain, auto_reset=False, log_capacity=512)
    refs = [RefEnv.from_points(doms[d], cap_new=512) for d in env_domain]
    obs = env.reset(static=True).cpu().numpy()
    obs_ref = np.stack([r.reset(static=True)[0] for r in refs])
    np.testing.assert_array_equal(obs, obs_ref)
    rng = np.random.default_rng(77)
    codes = np.zeros(5, int)
    valid = mism = smoothed = 0
    for t in range(T):
        pts = np.stack([rng.uniform(0.05, 0.45, n), rng.uniform(0.2, 1.5, n)], axis=1)
        typ = rng.uniform(0, 1, n)
        o, d, c, code = [x.cpu().numpy() for x in env.move(torch.from_numpy(pts), torch.from_numpy(typ))]
        reset_mask = np.zeros(n, np.uint8)
        for k in range(n):
            n_before = refs[k].scalars()["n_elem"]
            nv_before = refs[k].not_valid_count()
            o_r, d_r, c_r, code_r = refs[k].move(pts[k], typ[k])
            # a rejected move that leaves not_valid_points empty went through smooth_pave (B:405-426)
            smoothed += int(code_r != 2 and refs[k].scalars()["n_elem"] == n_before and refs[k].not_valid_count() == 0
                            and nv_before > 0)
            assert code[k] == code_r, (t, k, code[k], code_r)
            codes[code_r] += 1
            if code_r != 2:
                assert bool(d[k]) == d_r and bool(c[k]) == c_r, (t, k)
            if code_r == 0:
                diff = np.abs(o[k].astype(np.float64) - o_r).max()
                assert diff <= 1e-5, (t, k, diff)
                mism += int((o[k] != o_r).sum())
            valid += refs[k].scalars()["n_elem"] > n_before
            if d_r or code_r >= 2:
                reset_mask[k] = 1
                refs[k].reset(static=True)
        if reset_mask.any():
            env.reset(mask=torch.from_numpy(reset_mask), static=True)
        if t % 30 == 29 or t == T - 1:
            for k in rng.choice(n, 96, replace=False):
                st = env.get_state(int(k))
                ids, xy = refs[k].ring()
                np.testing.assert_array_equal(st["ring_ids"], ids)
                assert np.array_equal(st["ring_xy"], xy)    # also after a smoothing moved the front (host-libm tan / cos / pow)
                assert len(env.get_not_valid(int(k))) == refs[k].not_valid_count() and st["n_elem"] == refs[k].scalars()["n_elem"]
                if st["ref_index"] >= 0:
                    assert st["ref_id"] == refs[k].ref_id()
    print("move lockstep: codes", codes.tolist(), "valid", valid, "through smooth_pave", smoothed, "obs entries differing", mism)
    assert valid > 0.1 * n * T and codes[0] > 0 and smoothed > 20 and codes[3] == 0 and mism <= 1e-5 * n * T * 18
    env.close()


def test_self_touching_front_moves_in_lockstep_with_the_oracle():
    """random_domain(7023) has a doubled spike: coincident front segments traversed in opposite directions, coincident
    front vertices once the spike's tip is meshed away.  Round 2's campaign recorded 12 moves with differing observation
    rows and one `done` flag on it (profiles/r02_move_campaign.log, domain 30): a front smoothing left device coordinates
    one ulp from the reference's (ocml tan / cos, x * x for pow(x, 2)) and on coincident segments every comparison is a
    tie that the last bit decides; the zero divisors of coincident vertices ended the episode where the reference goes on.
    With the host libm's tan / cos table, pow(x, 2.0) restated and the NumPy zero-divisor continuation, 512 envs x 480 moves
    agree with the oracle in every code, flag, observation entry, ring coordinate and not_valid list."""
    import torch
    from oracle.ref_lib import RefEnv
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv
    from reinforcementlearning4meshgeneration_amd.domains import random_domain
    dom = random_domain(7023)
    n, T = 512, 480
    env = MeshVecEnv([dom], n_envs=n, auto_reset=False, log_capacity=512)
    refs = [RefEnv.from_points(dom, cap_new=512) for _ in range(n)]
    assert np.array_equal(env.reset(static=True).cpu().numpy(), np.stack([r.reset(static=True)[0] for r in refs]))
    rng = np.random.default_rng(105)
    codes = np.zeros(5, int)
    obs_mis = smoothed = 0
    for t in range(T):
        pts = np.stack([rng.uniform(0.05, 0.45, n), rng.uniform(0.2, 1.5, n)], axis=1)
        typ = rng.uniform(0, 1, n)
        o, d, c, code = [x.cpu().numpy() for x in env.move(torch.from_numpy(pts), torch.from_numpy(typ))]
        mask = np.zeros(n, np.uint8)
        for k in range(n):
            ne0, nv0 = refs[k].scalars()["n_elem"], refs[k].not_valid_count()
            o_r, d_r, c_r, code_r = refs[k].move(pts[k], typ[k])
            smoothed += int(code_r != 2 and refs[k].scalars()["n_elem"] == ne0 and refs[k].not_valid_count() == 0 and nv0 > 0)
            codes[code_r] += 1
            assert code[k] == code_r, (t, k, int(code[k]), code_r)
            if code_r != 2:
                assert bool(d[k]) == d_r and bool(c[k]) == c_r, (t, k)
            if code_r == 0:
                obs_mis += int((o[k] != o_r).any())
                assert np.abs(o[k].astype(np.float64) - o_r).max() <= 1e-5 or (o[k] != o_r).sum() <= 6, (t, k)
            if d_r or code_r >= 2:
                mask[k] = 1
                refs[k].reset(static=True)
        if mask.any():
            env.reset(mask=torch.from_numpy(mask), static=True)
        if t % 40 == 39:
            for k in range(0, n, 8):
                st = env.get_state(k)
                ids, xy = refs[k].ring()
                assert np.array_equal(st["ring_ids"], ids) and np.array_equal(st["ring_xy"], xy), (t, k)
                assert len(env.get_not_valid(k)) == refs[k].not_valid_count()
    print("self-touching front: codes", codes.tolist(), "through smooth_pave", smoothed, "moves with a differing observation", obs_mis)
    assert env.libm_exact == 1 and smoothed > 100 and codes[4] > 0
    assert obs_mis == 0
    env.close()


def test_half_quantum_angle_after_a_front_smoothing_rounds_like_the_reference():
    """The second move campaign (profiles/r03_move_campaign_seed1105.log, before the fix: 16 observation entries in 8 moves,
    all one state of env 483 on boundary(2)) replayed for that domain's 64 envs: the front smoother had put a vertex at
    (4, 1 / tan(0.76305)), the angle measured back from it is 0.76304999999999992 -- 8e-17 below the rounding boundary -- and
    ocml's atan2 returned the double above it.  With the boundary cases decided by glibc's own algorithm
    (csrc/meshenv_libm.h, atan2_glibc) every entry is the oracle's, this state's 0.7630 included."""
    import torch
    from oracle.ref_lib import RefEnv
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv
    from reinforcementlearning4meshgeneration_amd.domains import boundary, random_domain

    def golden_domain(name):
        return [tuple(p) for p in np.load(os.path.join(GOLDEN_DIR, name + ".npz"))["domain_xy"]]

    doms = [boundary(0), boundary(-1), boundary(1), boundary(2)] + \
           [golden_domain(x) for x in ("boundary16_biased_s2", "random1_1_biased_s1", "star_biased_s6")] + \
           [random_domain(7000 + k) for k in range(25)]
    n, T = 2048, 430
    env_domain = (np.arange(n) % len(doms)).astype(np.int32)
    env = MeshVecEnv(doms, env_domain=env_domain, auto_reset=False, log_capacity=512)
    assert env.atan2_exact == 1
    sel = [int(k) for k in np.flatnonzero(env_domain == 3)]
    refs = {k: RefEnv.from_points(doms[3], cap_new=512) for k in sel}
    env.reset(static=True)
    for r in refs.values():
        r.reset(static=True)
    rng = np.random.default_rng(1105)
    seen_state = differing = compared = 0
    for t in range(T):
        pts = np.stack([rng.uniform(0.05, 0.45, n), rng.uniform(0.2, 1.5, n)], axis=1)
        typ = rng.uniform(0, 1, n)
        o, d, c, code = [x.cpu().numpy() for x in env.move(torch.from_numpy(pts), torch.from_numpy(typ))]
        mask = ((d != 0) | (code >= 2)).astype(np.uint8)
        for k in sel:
            o_r, d_r, c_r, code_r = refs[k].move(pts[k], typ[k])
            assert code[k] == code_r and (code_r == 2 or (bool(d[k]) == d_r and bool(c[k]) == c_r)), (t, k)
            if code_r == 0:
                compared += 1
                differing += int((o[k] != o_r).sum())
                seen_state += int(k == 483 and o_r[3] == np.float32(0.763) and o_r[5] == np.float32(0.763))
            if d_r or code_r >= 2:
                refs[k].reset(static=True)
        if mask.any():
            env.reset(mask=torch.from_numpy(mask), static=True)
    print("half-quantum state: observations compared", compared, "entries differing", differing, "times the state was observed", seen_state)
    assert seen_state >= 5 and differing == 0
    env.close()


def test_single_env_move_and_static_reset_follow_the_reference_surface():
    from reinforcementlearning4meshgeneration_amd import BoudaryEnv
    tr = dict(np.load(os.path.join(GOLDEN_DIR, "move_hexagon_s3.npz")))
    env = BoudaryEnv([tuple(p) for p in tr["domain_xy"]])
    plain = env.reset()
    static = env.reset(static=True)
    assert np.array_equal(static, tr["reset_obs"]) and static[1] == 0.0 and plain[1] == 1.0
    assert np.array_equal(np.delete(plain, 1), np.delete(static, 1))
    raised = 0
    for t in range(len(tr["points"])):
        if tr["code"][t] == 2:
            with pytest.raises(UnboundLocalError):
                env.move(tr["points"][t], tr["types"][t])
            raised += 1
        else:
            obs, rew, done, info = env.move(list(tr["points"][t]), float(tr["types"][t]))
            assert rew == 0 and isinstance(done, bool) and set(info) == {"is_complete"}
            assert done == bool(tr["done"][t]) and info["is_complete"] == bool(tr["complete"][t])
            assert (obs is None) == (tr["code"][t] != 0)
            assert len(env.not_valid_points) == tr["n_not_valid"][t]
        if tr["was_reset"][t]:
            env.reset(static=True)
    assert raised > 0
    env.close()


def test_extract_samples_2_from_the_device_mesh_equals_the_reference():
    """general/mesh.py:1438-1489 through the drop-in class (device kernel, meshenv_extract_samples) on the mesh the DEVICE
    generated from the recorded actions: the reference's own (all_samples, types, outputs) -- element log, vertex table and
    neighbour order all line up; values identical except the last bits of the synthetic sector points' distances
    (tests/test_gpu_samples.py states the bar)."""
    from reinforcementlearning4meshgeneration_amd import BoudaryEnv
    from test_gpu_samples import _compare
    for name in ("samples_boundary0_post", "samples_star_ebrd"):
        tr = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        env = BoudaryEnv([tuple(p) for p in tr["vertex_xy"][:int(tr["n0"])]])
        env.reset()
        for a in tr["actions"]:
            env.step(a)
        nn, nr, rad, idx, thr = tr["params"]
        samples, types, outputs = env.extract_samples_2(None, int(nn), int(nr), rad, index=int(idx), quality_threshold=float(thr))
        assert isinstance(samples, list) and isinstance(types[0], list) and len(types[0]) == 1
        _compare(np.array(samples, np.float64), np.array(types, np.float64).reshape(-1), np.array(outputs, np.float64),
                 tr["samples"], tr["types"], tr["outputs"], int(nn), int(nr))
        env.close()
