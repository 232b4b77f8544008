"""GPU: BASELINE.json configs[2..4] at ONE GPU's share, in lockstep with the CPU oracle.

  configs[2]  4096 envs of ui/domains/boundary16.json (d1), SAC-shaped policy in the loop: FusedActor.sample ->
              MeshVecEnv.step, the oracle stepped with the very actions the actor emitted (closed loop)
  configs[3]  mixed d1/d2/d3 (120 / 196 / 272-vertex rings), 4096 envs = one of the 8 GPUs' shares
  configs[4]  GenerateRandomPolygon-style rings, 8192 envs = one GPU's share of 65536, ONE DOMAIN PER ENV
              (env_domain = arange(n): thousands of entries in the domain table, ragged rings)

Bars as everywhere: flags / topology / candidate order bit-exact, observation and reward within 1e-5; the oracle runs
OpenMP over envs so that each test stays within a minute or two on the box's 16 host cores."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from lockstep import run_lockstep

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    return torch


def _golden_domain(name):
    tr = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return [tuple(p) for p in tr["domain_xy"]]


def _biased(rng, T, n, frac=0.5):
    a = rng.uniform([-1, -1.5, 0], [1, 1.5, 1.5], size=(T, n, 3))
    pick = rng.random((T, n)) < frac
    b = np.stack([rng.uniform(-1, 1, (T, n)), rng.uniform(0.2, 1.0, (T, n)), rng.uniform(0.3, 1.2, (T, n))], axis=2)
    a[pick] = b[pick]
    return a.astype(np.float32)


def _ring_invariants(n0_of_env):
    """Size-independent properties of every sampled env (the ones of test_full_size_invariants_4096_envs that need no
    element log): clockwise ring, unique ids inside the vertex table, ring-length bookkeeping, area bookkeeping against
    the shoelace area of the current front (angles are 1e-4-rounded, hence the loose bound)."""
    def check(env, t):
        rng = np.random.default_rng(1000 + t)
        for k in rng.choice(env.num_envs, size=48, replace=False):
            st = env.get_state(int(k))
            if st["n"] <= 5 or st["n_elem"] == 0:
                continue
            n0 = n0_of_env[int(k)]
            n_new = st["n_vert"] - n0
            assert st["n"] == n0 - 2 * (st["n_elem"] - n_new), k
            xy = st["ring_xy"]
            shoelace = 0.5 * float(np.sum(xy[:, 0] * np.roll(xy[:, 1], -1) - np.roll(xy[:, 0], -1) * xy[:, 1]))
            orig = env.constants[st["domain"]].original_area
            assert shoelace < 0 and abs(-shoelace - st["current_area"]) <= 2e-3 * orig, k
            ids = st["ring_ids"]
            assert len(np.unique(ids)) == len(ids) and ids.min() >= 0 and ids.max() < st["n_vert"], k
    return check


def test_config3_sac_actor_in_the_loop_4096_d1_envs(torch_cuda):
    """configs[2]: the closed loop obs -> fused SAC actor (MFMA kernel, in-kernel Philox noise) -> meshenv_step on 4096
    boundary16 envs for 64 steps.  The oracle never sees the actor: it is stepped with the actions the actor produced
    from the DEVICE observations, so any observation drift would show up as diverging trajectories."""
    torch = torch_cuda
    from reinforcementlearning4meshgeneration_amd.actor import FusedActor
    torch.manual_seed(999)   # RL_Mesh.py: seed=999; random-init weights of the reference's architecture
    lin = [torch.nn.Linear(18, 128), torch.nn.Linear(128, 128), torch.nn.Linear(128, 128)]
    mu, ls = torch.nn.Linear(128, 3), torch.nn.Linear(128, 3)
    with torch.no_grad():   # spread the actions over the Box (a fresh SB3 actor sits near the centre, all rule 0)
        mu.weight.mul_(6.0)
        ls.bias.fill_(-0.5)
    for m in lin + [mu, ls]:
        m.cuda()
    actor = FusedActor.from_torch(lin, mu, ls)
    n, T = 4096, 64
    d1 = _golden_domain("boundary16_biased_s2")
    assert len(d1) == 120
    seen = []

    def policy(t, obs_dev):
        a = actor.sample(obs_dev, seed=999, counter=t)
        seen.append(a[:, 0].clone())
        return a

    st = run_lockstep(torch, [d1], np.zeros(n, np.int32), None, policy=policy, T=T, check_every=16, sample=64,
                      threads=16, invariants=_ring_invariants({k: 120 for k in range(n)}))
    rule = torch.stack(seen)
    frac = [float(((rule <= -0.5)).float().mean()), float(((rule > -0.5) & (rule < 0.5)).float().mean()),
            float((rule >= 0.5).float().mean())]
    print("config3 closed loop:", st, "rule mix", frac)
    assert min(frac) > 0.02            # all three rule types were exercised
    assert st["valid"] > 0.02 * n * T
    assert st["obs_mismatch"] <= 1e-6 * st["obs_total"]
    actor.close()


def test_config4_mixed_d1_d2_d3_one_gpu_share(torch_cuda):
    """configs[3] at one GPU's share: 4096 envs, env k on [boundary16, boundary15, test1][k % 3]."""
    doms = [_golden_domain(x) for x in ("boundary16_biased_s2", "boundary15_biased_s5", "test1_biased_s42")]
    assert [len(d) for d in doms] == [120, 196, 272]
    n, T = 4096, 48
    env_domain = (np.arange(n) % 3).astype(np.int32)
    a = _biased(np.random.default_rng(44), T, n, 0.6)
    st = run_lockstep(torch_cuda, doms, env_domain, a, check_every=16, sample=96, threads=16,
                      invariants=_ring_invariants({k: len(doms[k % 3]) for k in range(n)}))
    print("config4 mixed:", st)
    assert st["valid"] > 0.1 * n * T
    assert st["obs_mismatch"] <= 1e-6 * st["obs_total"]
    # sixteen 272-slot rings do not fit one CU's LDS; sixteen rings of their own lengths (120 / 196 / 272) do: the batch
    # steps with the CU-group kernel, its LDS packed by ring length (csrc/meshenv_kernels.h, GroupArgs::env_lds)
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv
    probe = MeshVecEnv(doms, env_domain=env_domain)
    assert probe.step_kernel == "meshenv::k_step_group<16, true, true, false>"
    probe.close()


def test_config5_one_random_domain_per_env_8192(torch_cuda):
    """configs[4] at one GPU's share: 8192 envs, each on its OWN random polygon (variable vertex count, ragged rings,
    an 8192-entry domain table initialised by a grid of 8192 k_init_domains waves)."""
    from reinforcementlearning4meshgeneration_amd.domains import random_domain
    n, T = 8192, 40
    doms = [random_domain(50_000 + k) for k in range(n)]
    sizes = np.array([len(d) for d in doms])
    assert sizes.min() >= 8 and len(np.unique(sizes)) > 20      # ragged
    env_domain = np.arange(n, dtype=np.int32)
    a = _biased(np.random.default_rng(55), T, n, 0.5)
    st = run_lockstep(torch_cuda, doms, env_domain, a, check_every=20, sample=128, threads=16,
                      invariants=_ring_invariants({k: int(sizes[k]) for k in range(n)}))
    print("config5 random:", st, "ring sizes", int(sizes.min()), int(sizes.max()))
    assert st["valid"] > 0.01 * n * T      # spiky star polygons accept few random actions (1.4 % here)
    assert st["obs_mismatch"] <= 1e-6 * st["obs_total"]


def test_config4_full_size_32768_mixed_envs_sampled_oracle_shadow(torch_cuda):
    """BASELINE.json configs[3] at its FULL size on one GPU (32 768 envs, env k on [d1, d2, d3][k % 3], the throughput
    kernel k_step<false>): 32 steps with 768 evenly spaced envs shadowed by the oracle on the same actions."""
    torch = torch_cuda
    from oracle.ref_lib import RefBatch, RefEnv
    from reinforcementlearning4meshgeneration_amd import MeshVecEnv
    doms = [_golden_domain(x) for x in ("boundary16_biased_s2", "boundary15_biased_s5", "test1_biased_s42")]
    n, T, S = 32768, 32, 768
    env_domain = (np.arange(n) % 3).astype(np.int32)
    env = MeshVecEnv(doms, env_domain=env_domain)
    assert env.step_kernel == "meshenv::k_step<false, true, false, false, true>"     # throughput form: early quad rejection
    pick = np.arange(0, n, n // S)[:S] + np.arange(S) % 3          # all three domains in the sample
    pick = np.unique(np.clip(pick, 0, n - 1))
    batch = RefBatch([RefEnv.from_points(doms[env_domain[k]], cap_new=64) for k in pick])
    assert np.array_equal(env.reset()[pick].cpu().numpy(), batch.reset())
    a_all = torch.from_numpy(_biased(np.random.default_rng(66), T, n, 0.6)).cuda()
    idx = torch.from_numpy(pick).cuda()
    for t in range(T):
        o, r, d, c = env.step(a_all[t])
        o_ref, r_ref, d_ref, c_ref = batch.step(a_all[t][idx].cpu().numpy(), auto_reset=True, threads=16)
        assert np.abs(o[idx].cpu().numpy().astype(np.float64) - o_ref).max() <= 1e-5, t
        assert np.abs(r[idx].cpu().numpy() - r_ref).max() <= 1e-5, t
        assert np.array_equal(d[idx].cpu().numpy(), d_ref) and np.array_equal(c[idx].cpu().numpy(), c_ref), t
    for j in range(0, len(pick), 12):
        st = env.get_state(int(pick[j]))
        ids, xy = batch.envs[j].ring()
        assert np.array_equal(st["ring_ids"], ids) and np.array_equal(st["ring_xy"], xy)
    cnt = env.counters()
    assert cnt["steps"] == n * T and cnt["valid"] > 0.1 * n * T
    env.close()
