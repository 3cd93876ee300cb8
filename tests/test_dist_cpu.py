"""Data-parallel plumbing on CPU: sharding rules and the collectives over gloo, world_size 2."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nvfpcc_amd import dist as nd


def test_minibatch_sharding_is_a_partition():
    order = np.random.default_rng(0).permutation(917)
    for world in (1, 2, 4, 8):
        B = 16
        for step in (0, 3, 57):   # step 57 is the short last batch (917 mod 16 = 5)
            parts = [nd.shard_minibatch(order, step, B, r, world) for r in range(world)]
            whole = parts[0][1]
            assert np.array_equal(whole, order[step * B:(step + 1) * B])
            merged = np.concatenate([p[0] for p in parts])
            assert sorted(merged.tolist()) == sorted(whole.tolist())
            assert max(len(p[0]) for p in parts) - min(len(p[0]) for p in parts) <= 1


def test_range_sharding_covers_everything_once():
    for n in (917, 4096, 5, 1):
        for world in (1, 2, 3, 8):
            spans = [nd.shard_range(n, r, world) for r in range(world)]
            covered = [i for lo, hi in spans for i in range(lo, hi)]
            assert covered == list(range(n))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = nd.init(backend="gloo")
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    nd.allreduce_sum_(flat)
    table = torch.zeros(7, 3, 2)
    lo, hi = nd.shard_range(7, rank, world)
    table[lo:hi] = rank + 1
    nd.allgather_rows_(table, rank, world)
    q.put((rank, flat.tolist(), table[:, 0, 0].tolist()))
    dist.destroy_process_group()


def test_gloo_allreduce_and_allgather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, flat, col in res:
        assert flat == [3.0 * i for i in range(10)]          # (1 + 2) * i
        assert col == [1.0, 1.0, 1.0, 1.0, 2.0, 2.0, 2.0]    # rows 0-3 from rank 0, 4-6 from rank 1
