"""GPU: MeshGeneration.extract_samples_2 on the device (meshenv_extract_samples, csrc/meshenv_samples.h) against the lists the
REFERENCE returned on meshes it generated itself (tests/golden/samples_*.npz) and against the host restatement on a batch.

Bar: the same samples in the same order; types and outputs (target distance / angle) bit-identical; sample entries
bit-identical except the distance entries of the synthetic sector points -- their coordinates pass through cos / sin of an
unquantised angle (ocml on the device, libm in the reference): <= 1e-14 relative there, counted."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu

NAMES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, "samples_*.npz")))


def _compare(samples, types, outputs, want_s, want_t, want_o, nn, nr):
    assert samples.shape == want_s.shape and len(types) == len(want_t) and outputs.shape == want_o.shape
    assert np.array_equal(types, want_t)
    assert np.array_equal(outputs, want_o)
    mm_dist = [2 * (nn + j) for j in range(nr)]                    # distance entries of the sector tuple
    rest = [c for c in range(samples.shape[1]) if c not in mm_dist]
    assert np.array_equal(samples[:, rest], want_s[:, rest])
    rel = np.abs(samples[:, mm_dist] - want_s[:, mm_dist]) / np.maximum(np.abs(want_s[:, mm_dist]), 1e-300)
    assert rel.max(initial=0.0) <= 1e-14, rel.max()
    return int((samples[:, mm_dist] != want_s[:, mm_dist]).sum()), samples[:, mm_dist].size


@pytest.mark.parametrize("name", NAMES)
def test_device_extract_samples_equals_the_reference_lists(name):
    import torch
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv
    tr = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    nn, nr, rad, idx, thr = tr["params"]
    n0 = int(tr["n0"])
    dom = [tuple(p) for p in tr["vertex_xy"][:n0]]
    env = MeshVecEnv([dom], n_envs=3, auto_reset=False, log_capacity=256)
    env.reset()
    for a in tr["actions"]:                                        # the mesh the reference extracted from, rebuilt on the device
        env.step(torch.from_numpy(np.tile(a, (3, 1))).cuda())
    q, v = env.get_elements(1)
    assert np.array_equal(q, tr["quads"]) and np.array_equal(v, tr["vertex_xy"])
    s, t, o, offs, st = env.extract_samples(int(nn), int(nr), float(rad), int(idx), float(thr))
    offs = offs.cpu().numpy()
    assert (st.cpu().numpy() == 0).all() and offs[1] - offs[0] == offs[2] - offs[1] == len(tr["samples"]) > 50
    for k in range(3):
        sl = slice(int(offs[k]), int(offs[k + 1]))
        diff, tot = _compare(s[sl].cpu().numpy(), t[sl].cpu().numpy(), o[sl].cpu().numpy(), tr["samples"], tr["types"], tr["outputs"],
                             int(nn), int(nr))
    print(f"{name}: {len(tr['samples'])} samples identical; synthetic-point distances differing in the last bits: {diff} of {tot}")
    env.close()


def test_device_extract_samples_batch_equals_host_restatement():
    """512 part-meshed envs on boundary(): every env's device samples against samples.extract_samples_2 on the same mesh (32
    envs checked), both parameter sets of the reference's callers; the archived episode (which = 'last') after a reset;
    masked-out envs report status 1 and contribute nothing."""
    import torch
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv, boundary
    from samples_host import extract_samples_2
    n = 512
    env = MeshVecEnv([boundary(0)], n_envs=n, auto_reset=False, log_capacity=128)
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    lo = torch.tensor([-1.0, 0.2, 0.3], device="cuda"); hi = torch.tensor([1.0, 1.0, 1.2], device="cuda")
    for _ in range(70):
        env.step((lo + (hi - lo) * torch.rand((n, 3), device="cuda", generator=g)).contiguous())
    total_checked = 0
    for (nn, nr, rad, idx, thr) in ((2, 3, 4.0, 1, 0.7), (3, 3, 6.0, 5, 0.7)):
        s, t, o, offs, st = env.extract_samples(nn, nr, rad, idx, thr)
        offs = offs.cpu().numpy(); s = s.cpu().numpy(); t = t.cpu().numpy(); o = o.cpu().numpy()
        assert (st.cpu().numpy() == 0).all() and offs[-1] > 1000
        for k in range(0, n, 16):
            q, v = env.get_elements(k)
            hs, ht, ho = extract_samples_2(q, v, 30, nn, nr, rad, index=idx, quality_threshold=thr)
            sl = slice(int(offs[k]), int(offs[k + 1]))
            assert len(hs) == offs[k + 1] - offs[k], (k, len(hs), offs[k + 1] - offs[k])
            if len(hs):
                _compare(s[sl], t[sl], o[sl], np.array(hs, np.float64), np.array(ht, np.float64).reshape(-1), np.array(ho, np.float64), nn, nr)
                total_checked += len(hs)
    assert total_checked > 2000
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda"); mask[::2] = 1
    s2, t2, o2, offs2, st2 = env.extract_samples(2, 3, 4.0, 1, 0.7, mask=mask)
    st2 = st2.cpu().numpy(); offs2 = offs2.cpu().numpy()
    assert (st2[1::2] == 1).all() and (st2[::2] == 0).all() and (np.diff(offs2)[1::2] == 0).all()
    # the archive: reset everything, the finished (here: interrupted) episodes are not archived -- an env must END to be archived
    before = env.extract_samples(2, 3, 4.0, 1, 0.7)[3].cpu().numpy()
    env2 = MeshVecEnv([boundary(0)], n_envs=64, auto_reset=True, log_capacity=128)
    env2.reset()
    for _ in range(400):
        env2.step((lo + (hi - lo) * torch.rand((64, 3), device="cuda", generator=g)).contiguous())
    sl, tl, ol, offl, stl = env2.extract_samples(2, 3, 4.0, 1, 0.7, which="last")
    offl = offl.cpu().numpy()
    checked = 0
    for k in range(64):
        le = env2.get_last_episode(k)
        if le["episodes"] == 0:
            continue
        hs, ht, ho = extract_samples_2(le["quads"], le["vertex_xy"], 30, 2, 3, 4.0, index=1, quality_threshold=0.7)
        assert len(hs) == offl[k + 1] - offl[k]
        checked += 1
    assert checked > 10 and before[-1] > 0
    env.close(); env2.close()
