"""Data-parallel plumbing on CPU: sharding rules and the collectives over gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
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


# ----------------------------------------------------------------------------------------------------------------
# host logic of the data-parallel epoch (engine.EpochDriver) and of bench.py's launcher, without a GPU
# ----------------------------------------------------------------------------------------------------------------
class _FakeEngine:
    """Stands in for TrainEngine: per-block 'gradients' are fixed vectors, the replicated term is scaled by
    rate_grad_scale -- exactly the structure dist.attach relies on."""

    def __init__(self, n, width=5):
        rng = np.random.default_rng(3)
        self.block_g = torch.from_numpy(rng.standard_normal((n, width)).astype(np.float32))
        self.replicated = torch.from_numpy(rng.standard_normal(width).astype(np.float32))
        self.counts = np.arange(1, n + 1, dtype=np.float64)
        self.rate_grad_scale, self.grad_hook, self.collective_mode = 1.0, None, None
        self.flat_g = torch.zeros(width)
        self.noise_step = self.opt_step = 0
        self.log = []

    def enable_epoch_stats(self):
        pass

    def train_step(self, ids, q, n_pts=None):
        self.noise_step += 1
        ids = np.asarray(ids, np.int64)
        self.flat_g = self.block_g[ids].sum(0) / n_pts + self.replicated * self.rate_grad_scale
        if self.grad_hook is not None:
            self.grad_hook(self.flat_g)
        self.opt_step += 1
        self.log.append((tuple(ids.tolist()), q, n_pts, self.flat_g.clone()))


def _epoch_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from nvfpcc_amd.engine import EpochDriver
    nd.init(backend="gloo")
    eng = _FakeEngine(13)
    nd.attach(eng, world)
    assert eng.rate_grad_scale == 1.0 / world and eng.collective_mode == "host"     # gloo cannot be graph-captured
    drv = EpochDriver(eng, 4, rank, world, use_graph=False)
    n = drv.run(np.random.default_rng(8).permutation(13), 1)
    q.put((rank, n, eng.noise_step, [(ids, npts, g.tolist()) for ids, _, npts, g in eng.log]))
    dist.destroy_process_group()


def test_epoch_driver_world2_short_and_empty_shares_reproduce_one_rank():
    """13 blocks, global mini-batch 4, two ranks over gloo: steps 0-2 give each rank two blocks, step 3 is ONE block
    (rank 1's share is empty).  After the all-reduce every rank holds the single-rank gradient of every step --
    block terms summed over ranks, the replicated term counted once -- and both ranks advanced their step counters
    identically (NVFPCC.py:149-223 under SURVEY.md 8(e))."""
    from nvfpcc_amd.engine import EpochDriver
    order = np.random.default_rng(8).permutation(13)
    ref = _FakeEngine(13)
    assert EpochDriver(ref, 4, 0, 1, use_graph=False).run(order, 1) == 4
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_epoch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n, noise_step, log in res:
        assert n == 4 and noise_step == 4
        for s, (ids, npts, g) in enumerate(log):
            whole = order[4 * s:4 * s + 4]
            assert list(ids) == list(whole[rank::2]) and npts == ref.log[s][2]
            np.testing.assert_allclose(g, ref.log[s][3].numpy(), rtol=1e-5, atol=1e-6)
    assert res[1][3][3][0] == () and len(res[0][3][3][0]) == 1          # the idle rank of the last step


class _FakeEvalEngine:
    """Stands in for TrainEngine in NVFPCC.test_log_fields: per-block additive sums (22 floats each), a replicated
    weight-rate term and point counts -- the structure the sharded evaluation relies on."""

    def __init__(self, n):
        rng = np.random.default_rng(21)
        self.N_leaf = n
        self.per_block = torch.from_numpy(rng.uniform(0.5, 2.0, size=(n, 22)))
        self.counts = rng.integers(800, 1100, size=n).astype(np.float64)

    def eval_sums(self, lo, hi, q=2):
        return self.per_block[lo:hi].sum(0) if hi > lo else torch.zeros(22, dtype=torch.float64)

    def weight_bits(self):
        return 12345.5


class _FakeNet:
    @staticmethod
    def get_network_bits():
        return 6789.0


def _eval_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import NVFPCC as cli
    nd.init(backend="gloo")
    eng = _FakeEvalEngine(13)
    f = cli.test_log_fields(eng, _FakeNet, 917 * 936.0, 200.0, rank, world, nd.allreduce_sum_)
    q.put((rank, [float(v) for v in f], cli.TEST_LINE % ((10, 0.0) + tuple(f))))
    dist.destroy_process_group()


def test_sharded_eval_world2_prints_the_one_rank_test_line():
    """The every-10th-epoch evaluation (NVFPCC.py:308-392) under data parallelism: each rank evaluates its contiguous shard
    (13 blocks on 2 ranks: 7 + 6) and one all-reduce of the 22 additive log sums gives BOTH ranks the full-batch fields;
    the formatted TEST line equals the one-rank line character for character."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import NVFPCC as cli
    ref = cli.test_log_fields(_FakeEvalEngine(13), _FakeNet, 917 * 936.0, 200.0)
    ref_line = cli.TEST_LINE % ((10, 0.0) + tuple(ref))
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, fields, line in res:
        np.testing.assert_allclose(fields, [float(v) for v in ref], rtol=1e-12)
        assert line == ref_line
    assert "b_all:" in ref_line and "PSNR1:" in ref_line


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment launches N ranks through torch.distributed.run
    (127.0.0.1 rendezvous) before touching the GPU; inside a torchrun environment it does not."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    calls = []

    class R:
        returncode = 0

    monkeypatch.setattr(bench.subprocess, "run", lambda cmd, **kw: calls.append((cmd, kw)) or R())
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    ran = []
    monkeypatch.setattr(bench, "run", lambda args: ran.append(args) or 0)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and not ran and len(calls) == 1
    cmd = calls[0][0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert calls[0][1]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit):
        bench.main()
    assert len(ran) == 1 and len(calls) == 1
