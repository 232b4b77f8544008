"""CPU: the C oracle (oracle/meshenv_ref.c) against the golden traces recorded from the reference itself
(oracle/gen_golden.py).  Bit-exact: observations, rewards, flags, ring topology, candidate list, new vertices."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_names
from oracle.ref_lib import RefEnv


def replay(tr, check_candidates=True):
    c = tr["consts"]
    env = RefEnv(tr["domain_xy"], c[0], c[2], c[3])
    obs, none = env.reset()
    assert not none
    np.testing.assert_array_equal(obs, tr["reset_obs"])
    ids, keys = env.candidates()
    np.testing.assert_array_equal(ids, tr["reset_cand_ids"])
    np.testing.assert_array_equal(keys, tr["reset_cand_keys"])
    assert env.ref_id() == int(tr["reset_ref_id"])
    n0 = tr["domain_xy"].shape[0]
    for t in range(len(tr["actions"])):
        obs, rew, done, comp, none = env.step(tr["actions"][t])
        assert none == bool(tr["obs_none"][t]), t
        if not none:
            np.testing.assert_array_equal(obs, tr["obs"][t], err_msg=f"obs step {t}")
        assert rew == tr["reward"][t], (t, rew, tr["reward"][t])
        assert done == bool(tr["done"][t]) and comp == bool(tr["complete"][t]), t
        rids, _ = env.ring()
        n = int(tr["ring_len"][t])
        assert len(rids) == n, t
        np.testing.assert_array_equal(rids, tr["ring_ids"][t, :n], err_msg=f"ring step {t}")
        if not none:
            assert env.ref_id() == tr["ref_id"][t], t
        sc = env.scalars()
        assert sc["n_elem"] == tr["n_elem"][t] and sc["failed_num"] == tr["failed_num"][t], t
        assert sc["current_area"] == tr["current_area"][t], t
        if check_candidates:
            ids, keys = env.candidates()
            m = min(int(tr["n_cand"][t]), n0)
            assert len(ids) == tr["n_cand"][t], t
            np.testing.assert_array_equal(ids[:m], tr["cand_ids"][t, :m], err_msg=f"cand ids step {t}")
            np.testing.assert_array_equal(keys[:m], tr["cand_keys"][t, :m], err_msg=f"cand keys step {t}")
        if not np.isnan(tr["new_xy"][t, 0]):
            _, vxy = env.elements()
            np.testing.assert_array_equal(vxy[sc["n_vert"] - 1], tr["new_xy"][t])
        if done and tr["auto_reset"]:
            env.reset()


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_trace(name):
    tr = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    replay(tr)


def test_golden_covers_edge_cases():
    """The fixture set must exercise the episode-control paths the reference has."""
    trs = [dict(np.load(os.path.join(GOLDEN_DIR, n + ".npz"))) for n in golden_names()]
    assert any(((t["done"] == 1) & (t["complete"] == 0)).any() for t in trs), "no 100-failure truncation"
    assert any(((t["done"] == 1) & (t["complete"] == 1)).any() for t in trs), "no completed episode"
    assert any(t["obs_none"].any() for t in trs), "no None observation"
    assert any((t["ring_len"] == 4).any() for t in trs) and any((t["ring_len"] == 5).any() for t in trs)
    assert any((~np.isnan(t["new_xy"][:, 0])).any() for t in trs)
    # accepted find_same_point (rl/boundary_env.py:165-175 -> general/mesh.py:623-629): a rule-0 action within 0.001 of a
    # ring vertex reuses the rule -1 quad -- a valid step with |a0| < 0.5 and no new vertex
    same = [int(((t["valid"] == 1) & (np.abs(t["actions"][:, 0]) < 0.5) & np.isnan(t["new_xy"][:, 0])).sum()) for t in trs]
    assert sum(same) >= 5 and sum(1 for s in same if s > 0) >= 2, same
    # a near-vertex point that is NOT accepted as well (hit, rule -1 quad invalid): boundary0_targeted steps 7 and 11
    assert all(int(t["valid"].sum()) > 0 or t["domain_xy"].shape[0] <= 5 for t in trs), "a fixture without any extraction"


# ------------------------------------------------------------------------------------------------ move() API (8f row 4)
from conftest import move_golden_names  # noqa: E402


def replay_move(tr, make_env, do_reset, do_move, state_of, obs_cmp=np.testing.assert_array_equal):
    """Shared by the oracle test (here) and the GPU test: replays a recorded move() trace through `do_move` and compares
    with what the reference returned.  codes: 0 ok, 1 obs None, 2 the reference raises UnboundLocalError (ring <= 5),
    4 it raises inside smooth_pave; calls that went through smooth_pave (B:405-426) are ordinary calls here."""
    env = make_env(tr)
    obs = do_reset(env)
    obs_cmp(obs, tr["reset_obs"])
    seen = np.zeros(5, int)
    for t in range(len(tr["points"])):
        obs, done, comp, code = do_move(env, tr["points"][t], float(tr["types"][t]))
        assert code == tr["code"][t], (t, code, tr["code"][t])
        seen[code] += 1
        if code == 0:
            obs_cmp(obs, tr["obs"][t], err_msg=f"obs move {t}")
        if code != 2:
            assert done == bool(tr["done"][t]) and comp == bool(tr["complete"][t]), t
        st = state_of(env)
        n = int(tr["ring_len"][t])
        assert st["n"] == n, t
        np.testing.assert_array_equal(st["ring_ids"], tr["ring_ids"][t, :n], err_msg=f"ring move {t}")
        assert st["n_elem"] == tr["n_elem"][t] and st["n_not_valid"] == tr["n_not_valid"][t], t
        if code == 0:
            assert st["ref_id"] == tr["ref_id"][t], t
        if not np.isnan(tr["new_xy"][t, 0]):
            np.testing.assert_array_equal(st["last_vertex"], tr["new_xy"][t])
        if tr["was_reset"][t]:
            do_reset(env)
    return seen


@pytest.mark.parametrize("name", move_golden_names())
def test_oracle_move_matches_reference_trace(name):
    tr = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))

    def state_of(env):
        ids, _ = env.ring()
        sc = env.scalars()
        _, vxy = env.elements()
        return dict(n=len(ids), ring_ids=ids, n_elem=sc["n_elem"], n_not_valid=env.not_valid_count(),
                    ref_id=env.ref_id(), last_vertex=vxy[sc["n_vert"] - 1])

    replay_move(tr, lambda tr: RefEnv(tr["domain_xy"], tr["consts"][0], tr["consts"][2], tr["consts"][3]),
                lambda env: env.reset(static=True)[0], lambda env, p, ty: env.move(p, ty), state_of)


def test_move_golden_covers_every_return_path():
    trs = [dict(np.load(os.path.join(GOLDEN_DIR, n + ".npz"))) for n in move_golden_names()]
    codes = np.sum([np.bincount(t["code"], minlength=5) for t in trs], axis=0)
    assert codes[0] > 0 and codes[2] > 0, codes     # ok / the reference raises on a finished ring
    smoothed = sum(int(t["smoothed"].sum()) for t in trs)
    assert smoothed >= 30                            # move() went through smooth_pave (B:405-426) ...
    assert any(((t["smoothed"] == 1) & (t["done"] == 1)).any() for t in trs)   # ... and ended an episode there (B:416-420)
    assert any(((t["done"] == 1) & (t["complete"] == 1)).any() for t in trs)      # ring of 4: complete
    assert any(((t["done"] == 1) & (t["complete"] == 0) & (t["code"] == 0)).any() for t in trs)  # ring of 5
    assert any((t["n_not_valid"] > 8).any() for t in trs)
    assert any((~np.isnan(t["new_xy"][:, 0])).any() for t in trs)
