"""CPU: tests/samples_host.extract_samples_2 (the host restatement the GPU tests compare the device kernel with) against the reference's MeshGeneration.extract_samples_2 (general/mesh.py:1438-1489)
on meshes the reference generated (tests/golden/samples_*.npz, oracle/gen_samples_golden.py): same samples in the same
order, value for value (pure Python float arithmetic on both sides)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from samples_host import extract_samples_2, segment_lists

NAMES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, "samples_*.npz")))


@pytest.mark.parametrize("name", NAMES)
def test_extract_samples_2_matches_the_reference(name):
    tr = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    nn, nr, rad, idx, thr = tr["params"]
    samples, types, outputs = extract_samples_2(tr["quads"], tr["vertex_xy"], int(tr["n0"]), int(nn), int(nr), rad, index=int(idx),
                                                quality_threshold=float(thr))
    assert len(samples) == len(tr["samples"]) > 50
    assert np.array_equal(np.array(samples, np.float64), tr["samples"])
    assert np.array_equal(np.array(types, np.float64).reshape(-1), tr["types"])
    assert np.array_equal(np.array(outputs, np.float64), tr["outputs"])
    assert set(np.unique(tr["types"])) == {0.0, 0.5, 1.0}


def test_segment_lists_order_follows_connect_vertices():
    # one quad on a square ring [0, 1, 2, 3] plus a generated vertex 4: element [4, 0, 1, 2] (new vertex first, as B:177-182)
    adj = segment_lists(np.array([[4, 0, 1, 2]]), 5, 4)
    assert adj[0] == [3, 1, 4] and adj[4] == [2, 0] and adj[2] == [1, 3, 4] and adj[1] == [0, 2]
