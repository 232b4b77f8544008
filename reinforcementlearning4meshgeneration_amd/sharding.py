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


class BucketExchange:
    """The per-step exchange of the multi-GPU path, off the per-step critical path.

    The step kernel writes the [n, 21] message of step t straight into slot ``t % steps`` of a ``[steps, n, 21]``
    bucket (``slot(t)`` is the tensor to hand to ``MeshVecEnv.set_packed_output`` before launching step t); when the
    last slot of a bucket has been launched, ``after_step(t)`` issues ONE ``all_gather_into_tensor`` for the whole
    bucket (asynchronous: RCCL runs it on its own stream, ordered after the launch stream's work at the time of the
    call) and the next bucket's steps proceed meanwhile.  Two buckets: bucket b is rewritten only after ``wait()`` on the
    collective that last read it (a stream-level wait for RCCL, a host wait for gloo).  ``consumer(gathered, t0)``, if
    given, is called with the rank-major ``[world * steps, n, 21]`` result of the bucket whose first step is t0 once
    that collective has been waited for -- the learner's replay-buffer append.

    ``stage_to_cpu``: the gloo rehearsal on a GPU box (gloo cannot read device memory): the bucket is copied to the host
    and gathered synchronously.
    """

    def __init__(self, dist, torch, n_envs: int, steps: int, device, stage_to_cpu: bool = False, consumer=None,
                 group=None):
        self.dist, self.torch, self.group = dist, torch, group
        self.steps = max(1, int(steps))
        self.world = dist.get_world_size(group)
        self.stage_to_cpu = bool(stage_to_cpu)
        self.consumer = consumer
        gdev = torch.device("cpu") if stage_to_cpu else device
        self.buckets = [torch.zeros((self.steps, n_envs, MSG_DIM), dtype=torch.float32, device=device) for _ in range(2)]
        self.gathered = [torch.empty((self.world * self.steps, n_envs, MSG_DIM), dtype=torch.float32, device=gdev)
                         for _ in range(2)]
        # device addresses of every slot, for callers that hand the pointer to the C-ABI themselves (bench.py's timed loop)
        self._slot_ptr = [[int(self.buckets[b][k].data_ptr()) for k in range(self.steps)] for b in range(2)]
        self.works = [None, None]      # in-flight collective per bucket
        self.first_step = [None, None]  # first step of the bucket contents handed to that collective
        self.collectives = 0           # collectives issued
        self.steps_sent = 0            # step messages handed to a collective
        self._open = None              # (bucket, slots written, first step) of the bucket being filled
        self._result = [None, None]    # view of gathered[b] the in-flight collective writes

    def _bucket(self, t: int) -> int:
        return (t // self.steps) & 1

    def _retire(self, b: int):
        if self.works[b] is not None:
            self.works[b].wait()
            self.works[b] = None
        if self.first_step[b] is not None:
            if self.consumer is not None:
                self.consumer(self._result[b], self.first_step[b])
            self.first_step[b] = None

    def slot(self, t: int):
        """Message buffer of step t.  Entering a bucket first retires the collective that last read it."""
        b, k = self._bucket(t), t % self.steps
        if k == 0:
            self._retire(b)
        return self.buckets[b][k]

    def slot_ptr(self, t: int) -> int:
        """slot(t) as a raw device address (same retire-before-reuse rule)."""
        b, k = self._bucket(t), t % self.steps
        if k == 0:
            self._retire(b)
        return self._slot_ptr[b][k]

    def _send(self, b: int, m: int, t0: int):
        """One collective for the first m slots of bucket b (m == steps: the whole bucket)."""
        src = self.buckets[b] if m == self.steps else self.buckets[b][:m]
        dst = self.gathered[b] if m == self.steps else self.gathered[b].view(-1)[: self.world * src.numel()].view(
            self.world * m, *src.shape[1:])
        if self.stage_to_cpu:
            self.dist.all_gather_into_tensor(dst, src.cpu(), group=self.group)
        else:
            self.works[b] = self.dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True)
        self.first_step[b] = t0
        self._result[b] = dst
        self.collectives += 1
        self.steps_sent += m

    def after_step(self, t: int):
        b, k = self._bucket(t), t % self.steps
        self._open = (b, k + 1, t - k)
        if k != self.steps - 1:
            return
        self._send(b, self.steps, t - k)
        self._open = None

    def flush(self):
        """Send the partially filled bucket, if there is one (the consumer then receives ``[world * m, n, 21]`` with
        m < steps).  A run whose length is not a multiple of ``steps`` ends with this call, so that every step's message
        is gathered; afterwards ``drain()`` and the next step index may start again at 0."""
        if self._open is not None:
            b, m, t0 = self._open
            self._send(b, m, t0)
            self._open = None

    def drain(self):
        """Wait for everything in flight (oldest bucket first).  A partially filled bucket is not sent (``flush()``
        first if it should be)."""
        order = sorted((b for b in (0, 1) if self.first_step[b] is not None), key=lambda b: self.first_step[b])
        for b in order:
            self._retire(b)
        self._open = None
