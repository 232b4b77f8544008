"""CPU: the N > 1 exchange path (pack -> all_gather_into_tensor -> unpack) on world_size 2 with gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reinforcementlearning4meshgeneration_amd import sharding
    lo, hi = sharding.shard_range(n_total, world, rank)
    n = hi - lo
    ids = torch.arange(lo, hi, dtype=torch.float32)
    obs = ids[:, None] * 100 + torch.arange(18, dtype=torch.float32)[None, :]
    reward = ids.double() * 0.5
    done = (ids.long() % 3 == 0).to(torch.uint8)
    comp = (ids.long() % 2 == 0).to(torch.uint8)
    msg = sharding.pack_message(torch, obs, reward, done, comp)
    assert msg.shape == (n, sharding.MSG_DIM)
    gathered = sharding.gather_messages(dist, msg)
    o, r, d, c = sharding.unpack_message(gathered)
    all_ids = torch.arange(n_total, dtype=torch.float32)
    ok = bool(torch.equal(o, all_ids[:, None] * 100 + torch.arange(18, dtype=torch.float32)[None, :])
              and torch.equal(r, all_ids * 0.5) and torch.equal(d, all_ids.long() % 3 == 0)
              and torch.equal(c, all_ids.long() % 2 == 0))
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t.item()) == float(world)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world_size_2():
    world, n_total = 2, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
