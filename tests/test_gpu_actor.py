"""GPU: the fused SAC actor kernel against a plain PyTorch fp32 forward of the same network (floating-point
kernel -> torch reference; tolerance 2e-5 absolute on actions of magnitude <= 1.5: fp32 dot products of length 128
summed in a different order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _net(torch, seed):
    torch.manual_seed(seed)
    lin = [torch.nn.Linear(18, 128), torch.nn.Linear(128, 128), torch.nn.Linear(128, 128)]
    mu, ls = torch.nn.Linear(128, 3), torch.nn.Linear(128, 3)
    for m in lin + [mu, ls]:
        m.cuda()
    return lin, mu, ls


@pytest.mark.parametrize("n", [1, 31, 32, 4096, 5000])
def test_fused_actor_matches_torch(n):
    import torch

    from reinforcementlearning4meshgeneration_amd.actor import FusedActor
    from reinforcementlearning4meshgeneration_amd.vec_env import ACTION_HIGH, ACTION_LOW
    lin, mu, ls = _net(torch, 999)
    actor = FusedActor.from_torch(lin, mu, ls)
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    obs = (torch.rand((n, 18), device="cuda", generator=g) * 4 - 1).float()      # observation-like magnitudes
    noise = torch.randn((n, 3), device="cuda", generator=g)
    low = torch.as_tensor(ACTION_LOW, device="cuda")
    high = torch.as_tensor(ACTION_HIGH, device="cuda")
    with torch.no_grad():
        h = obs
        for m in lin:
            h = torch.relu(m(h))
        m_, s_ = mu(h), ls(h).clamp(-20, 2).exp()
        ref_det = low + 0.5 * (torch.tanh(m_) + 1) * (high - low)
        ref_sto = low + 0.5 * (torch.tanh(m_ + s_ * noise) + 1) * (high - low)
    det = actor.forward(obs)
    sto = actor.forward(obs, noise)
    assert det.shape == (n, 3) and det.dtype == torch.float32
    assert float((det - ref_det).abs().max()) <= 2e-5
    assert float((sto - ref_sto).abs().max()) <= 2e-5
    assert bool(((det >= low) & (det <= high)).all())
    actor.close()
