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


# ---------------------------------------------------------------------------------------------------------------
# bench.py's N > 1 schedule: sharding.BucketExchange (two [GS, n, 21] buckets, one async all_gather per bucket) driven
# by a fake step kernel that writes a recognisable message into the slot it is handed.

def _bucket_worker(rank, world, port, n, GS, T, flush, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reinforcementlearning4meshgeneration_amd import sharding

    def message(r, t):  # what rank r's step kernel writes at step t
        base = torch.arange(n, dtype=torch.float32)[:, None] * 1000 + torch.arange(sharding.MSG_DIM, dtype=torch.float32)[None, :]
        return base + 1e6 * r + 1e5 * (t % 7) + t

    seen, errors = [], []

    def consumer(gathered, t0):
        seen.append(t0)
        m = gathered.shape[0] // world       # GS, or fewer for a flushed partial bucket
        g = gathered.view(world, m, n, sharding.MSG_DIM)
        for r in range(world):
            for k in range(m):
                if not torch.equal(g[r, k], message(r, t0 + k)):
                    errors.append((t0, r, k))

    xch = sharding.BucketExchange(dist, torch, n, GS, torch.device("cpu"), consumer=consumer)
    in_flight_max = 0
    for t in range(T):
        slot = xch.slot(t)
        # a bucket may only be rewritten once the collective that read it has been retired
        assert xch.works[(t // GS) & 1] is None and xch.first_step[(t // GS) & 1] is None
        slot.copy_(message(rank, t))        # the "kernel"
        xch.after_step(t)
        in_flight_max = max(in_flight_max, sum(w is not None for w in xch.works))
    if flush:
        xch.flush()
        in_flight_max = max(in_flight_max, sum(w is not None for w in xch.works))
    xch.drain()
    full = T // GS
    sent = full + (1 if flush and T % GS else 0)
    ok = (not errors and seen == [GS * i for i in range(sent)] and xch.collectives == sent
          and xch.steps_sent == (T if flush else full * GS)
          and all(w is None for w in xch.works) and in_flight_max >= 1)
    q.put((rank, ok, len(errors), seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("GS,T,flush", [(8, 40, False), (3, 20, False), (1, 5, False), (3, 20, True), (32, 20, True), (8, 40, True)])
def test_bucket_exchange_schedule_world_size_2(GS, T, flush):
    """Every gathered bucket holds exactly what both ranks wrote, in step order; buckets are consumed oldest first; no
    bucket is handed out again before its collective was waited for; a trailing partial bucket is sent by flush() and
    only by flush()."""
    world, n = 2, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, n, GS, T, flush, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, n_err, seen in results:
        assert ok, (rank, n_err, seen)


# ---------------------------------------------------------------------------------------------------------------
# bench.py's own region loop (bench.run_region + bench.gather_steps) under the driver's arguments, with a fake kernel.

def _region_worker(rank, world, port, n, K, W, gather_every, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes
    import importlib.util
    from reinforcementlearning4meshgeneration_amd import sharding
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    GS = bench.gather_steps(gather_every, K)
    got = {}

    def consumer(gathered, j0):
        m = gathered.shape[0] // world
        g = gathered.view(world, m, n, sharding.MSG_DIM)
        for r in range(world):
            for k in range(m):
                got[(phase[0], r, j0 + k)] = float(g[r, k, 0, 0])

    xch = sharding.BucketExchange(dist, torch, n, GS, torch.device("cpu"), consumer=consumer)
    phase = ["warmup"]

    def launch(t, slot_ptr):   # the "kernel": writes 1000 * rank + absolute action index into the slot it was handed
        nbytes = n * sharding.MSG_DIM * 4
        buf = (ctypes.c_float * (n * sharding.MSG_DIM)).from_address(slot_ptr)
        val = 1000.0 * rank + t
        for i in range(n * sharding.MSG_DIM):
            buf[i] = val
        assert nbytes > 0

    cw, sw = bench.run_region(xch, launch, 0, W)
    phase[0] = "timed"
    ck, sk = bench.run_region(xch, launch, W, K)
    exp = {("timed", r, j): 1000.0 * r + W + j for r in range(world) for j in range(K)}
    exp.update({("warmup", r, j): 1000.0 * r + j for r in range(world) for j in range(W)})
    ok = (got == exp and sk == K and sw == W and ck == -(-K // GS) and ck >= 1 and (W == 0 or cw >= 1)
          and all(w is None for w in xch.works))
    q.put((rank, ok, ck, sk, GS))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K,W,gather_every", [(20, 5, 32), (200, 20, 32), (1, 0, 32), (33, 3, 8)])
def test_bench_region_gathers_every_step_inside_the_timed_region(K, W, gather_every):
    """The driver runs --steps 20 --warmup 5 against a default bucket of 32 steps: round 2's loop then issued no
    collective at all.  The region loop clamps the bucket to the region and flushes the partial one."""
    world, n = 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_region_worker, args=(r, world, port, n, K, W, gather_every, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ck, sk, GS in results:
        assert ok, (rank, ck, sk, GS)
        assert GS <= K
