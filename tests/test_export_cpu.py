"""CPU: the .inp writer against the file the reference itself wrote (fixture), fed with the oracle's elements."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR


def test_inp_writer_reproduces_reference_file(tmp_path):
    from oracle.ref_lib import RefEnv
    from reinforcementlearning4meshgeneration_amd import boundary
    from reinforcementlearning4meshgeneration_amd.export import inp_text, write_inp
    fx = json.load(open(os.path.join(GOLDEN_DIR, "inp_boundary0_uniform_s7.json")))
    tr = np.load(os.path.join(GOLDEN_DIR, fx["trace"] + ".npz"))
    env = RefEnv.from_points(boundary(0))
    env.reset()
    for t in range(fx["step"] + 1):
        _, _, done, comp, _ = env.step(tr["actions"][t])
        if t < fx["step"] and done:
            env.reset()
    assert done and comp
    quads, vxy = env.elements()
    assert len(quads) == fx["n_elements"]
    assert inp_text(quads, vxy, boundary(0)) == fx["inp"]
    out = tmp_path / "mesh.inp"
    write_inp(out, quads, vxy, boundary(0))
    assert out.read_text() == fx["inp"]
    with pytest.raises(ValueError):
        inp_text(quads[:3], vxy, boundary(0))       # unfinished mesh: boundary vertices without an element
    with pytest.raises(ValueError):
        inp_text(quads[:0], vxy, boundary(0))


def test_write_2_file_reproduces_reference_json(tmp_path):
    """mesh_graph / write_2_file against the JSON the reference's write_2_file produced (fixture)."""
    from oracle.ref_lib import RefEnv
    from reinforcementlearning4meshgeneration_amd import boundary
    from reinforcementlearning4meshgeneration_amd.export import write_2_file
    fx = json.load(open(os.path.join(GOLDEN_DIR, "write2file_boundary0_biased_s1.json")))
    tr = np.load(os.path.join(GOLDEN_DIR, fx["trace"] + ".npz"))
    env = RefEnv.from_points(boundary(0))
    env.reset()
    for t in range(fx["step"] + 1):
        _, _, done, comp, _ = env.step(tr["actions"][t])
        if t < fx["step"] and done:
            env.reset()
    assert done and comp
    quads, vxy = env.elements()
    out = tmp_path / "mesh.json"
    write_2_file(out, quads, vxy, boundary(0))
    assert json.load(open(out)) == fx["json"]


def test_oracle_element_quality_matches_reference_records():
    """The oracle's per-element measures are bit-identical to the reference's Mesh methods (fixture recorded by
    oracle/gen_golden.py: 219 elements of 8 episodes on three domains); statistics against numpy."""
    from oracle.ref_lib import element_quality, quality_stats
    z = np.load(os.path.join(GOLDEN_DIR, "quality_quads.npz"))
    rec = element_quality(z["quad_xy"])
    np.testing.assert_array_equal(rec, z["expected"])
    for ep in np.unique(z["episode"]):
        r = rec[z["episode"] == ep]
        st = quality_stats(r)
        np.testing.assert_array_equal(st[:, 0], r.min(0))
        np.testing.assert_array_equal(st[:, 2], r.max(0))
        np.testing.assert_allclose(st[:, 1], r.mean(0), rtol=1e-13)
        np.testing.assert_allclose(st[:, 3], r.var(0), rtol=1e-9, atol=1e-12)
