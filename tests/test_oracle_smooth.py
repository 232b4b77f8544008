"""CPU: the oracle's smooth_pave(interior=True) against records of the reference's own (tests/golden/smooth_*.npz,
written by oracle/gen_golden.py --smooth-only from general/mesh.py:790-795,1258-1288): vertex table after every call
BIT-exact, sweep count, rebuilt candidate list, and every step() after it (the smoothed state is the one stepped on)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, final_smooth_golden_names, front_smooth_golden_names, smooth_golden_names
from smooth_replay import replay, replay_final, replay_front


class OracleImpl:
    def __init__(self, tr):
        from oracle.ref_lib import RefEnv
        c = tr["consts"]
        self.env = RefEnv(tr["domain_xy"], c[0], c[2], c[3], cap_new=512)

    def reset(self):
        return self.env.reset()[0]

    def step(self, a):
        o, r, d, c, _none = self.env.step(a)
        return o, r, d, c

    def smooth(self, iteration):
        return self.env.smooth_interior(iteration)[0]

    def smooth_full(self, iteration):
        return self.env.smooth_pave_full(iteration)

    def ref_id(self):
        return self.env.ref_id()

    def smooth_final(self, iteration):
        return self.env.smooth_final(iteration)[0]

    def vertices(self):
        return self.env.elements()[1]

    def elements(self):
        return self.env.elements()[0]

    def ring_ids(self):
        return self.env.ring()[0]

    def candidates(self):
        return self.env.candidates()


@pytest.mark.parametrize("name", smooth_golden_names())
def test_oracle_smooth_interior_matches_reference_records(name):
    tr = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    moved = replay(tr, OracleImpl(tr))
    assert int(tr["n_calls"]) >= 8 and moved > 0


def test_smooth_fixtures_cover_both_stop_rules():
    trs = [dict(np.load(os.path.join(GOLDEN_DIR, n + ".npz"))) for n in smooth_golden_names()]
    assert any((t["call_sweeps"] == int(t["iteration"])).any() for t in trs)       # stopped by the iteration cap
    assert any(((t["call_sweeps"] > 1) & (t["call_sweeps"] < int(t["iteration"]))).any() for t in trs)   # by the 0.001 rule
    assert any((t["call_sweeps"] >= 40).any() for t in trs)
    assert any((t["call_nv"] - t["domain_xy"].shape[0] >= 30).any() for t in trs)   # dozens of generated vertices


@pytest.mark.parametrize("name", final_smooth_golden_names())
def test_oracle_smooth_of_finished_meshes_matches_reference_records(name):
    """MeshGeneration.smooth (general/mesh.py:1290-1392) on episodes that ended complete: same sweep counts, vertex table
    bit-identical (the two estimate branches call atan2 / cos / sin: the same libm as the recording interpreter here)."""
    tr = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    assert int(tr["n_calls"]) >= 3
    replay_final(tr, OracleImpl(tr), vertex_tol=0.0)


def test_final_smooth_fixtures_cover_all_three_branches_and_both_front_sizes():
    trs = [dict(np.load(os.path.join(GOLDEN_DIR, n + ".npz"))) for n in final_smooth_golden_names()]
    visits = np.sum([t["call_branch"].sum(axis=0) for t in trs], axis=0)
    assert (visits > 50).all(), visits                    # 1 / 2 / >= 3 related elements
    fronts = np.concatenate([t["call_nr"] for t in trs])
    assert (fronts == 4).any() and (fronts == 5).any()
    assert any((t["call_sweeps"] == int(t["iteration"])).any() for t in trs)


@pytest.mark.parametrize("name", front_smooth_golden_names())
def test_oracle_front_smoother_matches_reference_records(name):
    """smooth_pave(..., interior=False) -- smooth_current_boundary_3 (general/mesh.py:939-1028) + the interior relaxation +
    the rebuild -- and the find_next_state after it: vertex table, sweeps, observation, reference vertex, candidate list
    and every later step() as the reference recorded them."""
    tr = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    worst, front_moves = replay_front(tr, OracleImpl(tr))
    assert int(tr["n_calls"]) >= 5 and front_moves > 0


def test_front_constructions_match_the_reference_known_answers():
    """middle_vertex / side_vertex / indention_vertex (general/mesh.py:805-909) and Mesh.estimate_4th_vertex: 16 000
    evaluations by the reference itself
    (random, axis-aligned -- the B == 0 / A == 0 branches -- and grid inputs; oracle/gen_construction_golden.py), the
    oracle's bit for bit, including the inputs on which the reference raises."""
    from oracle.ref_lib import lib
    L = lib()
    tr = np.load(os.path.join(GOLDEN_DIR, "front_constructions.npz"))
    out = np.zeros(2, np.float64)
    seen_raise = 0
    for row, want, bad in zip(tr["inputs"], tr["outputs"], tr["raised"]):
        rc = L.meshenv_ref_front_construction(int(row[0]), np.ascontiguousarray(row[1:9]), out)
        assert rc == int(bad), (row, rc, bad)
        if not bad:
            assert np.array_equal(out, want), (row, out, want)
        seen_raise += int(bad)
    assert seen_raise > 10
