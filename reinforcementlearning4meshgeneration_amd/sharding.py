"""Multi-GPU layout of the environment batch: one process per GPU, contiguous env shards, no data-path
collective inside the step; the only exchange is the gather of the per-step (obs | reward | done | complete)
message for a central learner.  `torch.distributed` backend "nccl" is RCCL on ROCm; the same
code runs over "gloo" on CPU tensors, which is how the tests exercise it."""
from __future__ import annotations

from typing import List, Sequence, Tuple

MSG_DIM = 21  # 18 obs + reward + done + complete, float32


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of environments owned by `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(int(n_total), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_domains(env_domain: Sequence[int], world: int, rank: int) -> List[int]:
    """Domain index of every env in this rank's shard.  Mixed-domain batches should interleave domains
    (env k -> domain k % D) so that each shard sees the same mix and the same total ring length."""
    lo, hi = shard_range(len(env_domain), world, rank)
    return list(env_domain[lo:hi])


def pack_message(torch, obs, reward, done, complete, out=None):
    """[n, 21] float32 message: obs | reward | done | complete (one fused buffer -> one collective)."""
    n = obs.shape[0]
    if out is None:
        out = torch.empty((n, MSG_DIM), dtype=torch.float32, device=obs.device)
    out[:, :18] = obs
    out[:, 18] = reward
    out[:, 19] = done
    out[:, 20] = complete
    return out


def unpack_message(msg):
    return msg[:, :18], msg[:, 18], msg[:, 19] != 0, msg[:, 20] != 0


def gather_messages(dist, msg, gathered=None, group=None):
    """all_gather_into_tensor of equal-sized shards (pad the last shard if sizes differ)."""
    import torch
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world * msg.shape[0], msg.shape[1]), dtype=msg.dtype, device=msg.device)
    dist.all_gather_into_tensor(gathered, msg, group=group)
    return gathered
