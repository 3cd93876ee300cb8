"""Data parallelism over leaf blocks: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-device (NVFPCC.py:105); this is what the build adds.
Leaf blocks are independent given the shared decoder and every loss term is a SUM over blocks, so

  * mini-batch phase: position j of the epoch order goes to rank j mod W; after backward ONE
    all-reduce (SUM) of the flat fp32 gradient buffer (52 219 floats = 209 KB at chanstr 8,16,8,8 --
    latency-bound on xGMI, far below the per-link bandwidth regime), then every rank applies the same
    fused Adam update to its replica;
  * the weight-rate term is identical on every rank, so its gradient is scaled by 1/W before the SUM -- on EVERY
    rank, including one whose share of a short last mini-batch is empty (TrainEngine._idle_backward), so the summed
    gradient equals the single-GPU one whatever W is;
  * n_pts (points of the global mini-batch) is computed on the host from per-block counts: no collective;
  * weight noise (q = 1) is keyed by (seed, step, layer) and so identical on all ranks; latent noise is
    keyed by (block id, step), so results do not depend on W; the step counter advances once per mini-batch and
    once per latent phase on every rank, idle or not;
  * the all-reduce is launched from the host behind the captured step graph (engine.GraphedTrainStep;
    torch.distributed runs RCCL on a stream of its own: two event waits per step).  NVF_GRAPH_COLLECTIVE=graph opts in
    to capturing it as a node of the step graph (and of the unrolled 16/8/4/2-step graphs): measured with a one-rank
    group only, so it stays opt-in until a >= 2-rank RCCL run has passed with it;
  * latent phase / eval: contiguous block shards, no collective inside the step; one all-gather of the
    updated latent rows per epoch.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for one process)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # NVF_DIST_BACKEND=gloo is a test hook (several ranks sharing one GPU); production is RCCL
            backend = os.environ.get("NVF_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_minibatch(order, step, batch, rank, world):
    """Block ids of mini-batch `step` owned by `rank`: position j of the epoch order -> rank j mod W.
    Returns (ids of this rank, ids of the whole mini-batch)."""
    whole = np.asarray(order[step * batch:(step + 1) * batch], np.int64)
    return whole[rank::world], whole


def shard_range(n, rank, world):
    """Static contiguous shard [lo, hi) of n blocks for the latent / eval / decode phases."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def allreduce_sum_(flat):
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and flat.is_cuda:     # test hook: gloo reduces on the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allgather_rows_(table, rank, world):
    """Re-synchronise a replicated [N, ...] table after every rank updated its shard_range rows."""
    if not (dist.is_initialized() and world > 1):
        return table
    n = table.shape[0]
    per = (n + world - 1) // world
    width = table[0].numel()
    pad = torch.zeros(per * world, width, device=table.device, dtype=table.dtype)
    lo, hi = shard_range(n, rank, world)
    mine = torch.zeros(per, width, device=table.device, dtype=table.dtype)
    mine[:hi - lo] = table[lo:hi].reshape(hi - lo, width)
    if dist.get_backend() == "gloo" and table.is_cuda:        # test hook: gloo gathers on the host
        parts = [torch.zeros(per, width) for _ in range(world)]
        dist.all_gather(parts, mine.cpu())
        pad.copy_(torch.cat(parts, 0))
    elif hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(pad, mine)
    else:
        dist.all_gather(list(pad.chunk(world)), mine)
    table.copy_(pad[:n].reshape(table.shape))
    return table


def attach(engine, world, force=False):
    """Wire an engine for data parallelism.  ``force``: install the all-reduce hook even for one rank (exercises the
    RCCL plumbing -- communicator, capture inside the step graph -- on a single GPU).

    Where the all-reduce sits (engine.GraphedTrainStep): "host" (default) ends the graph after the backward pass and
    launches the all-reduce and the optimiser node from the host; NVF_GRAPH_COLLECTIVE=graph captures it as a node of
    the step graph (the hand-over between the compute stream and RCCL's stream is then a graph edge; measured 0.5113 vs
    0.522 ms per step, ONE-rank group on one MI355X -- never run with peers, hence opt-in).  RCCL builds its communicator
    on the FIRST collective and doing that inside a capture invalidates it, so one eager all-reduce is issued here --
    by every rank, once, whether or not a rank will ever build a graph (a rank whose share of every full mini-batch
    is empty never does)."""
    if world > 1 or (force and dist.is_initialized()):
        engine.rate_grad_scale = 1.0 / world
        engine.grad_hook = _allreduce_any_world if force and world == 1 else allreduce_sum_
        # gloo (test hook) reduces through host memory, which cannot be captured into a HIP graph
        if dist.get_backend() == "gloo":
            engine.collective_mode = "host"
        else:
            engine.collective_mode = os.environ.get("NVF_GRAPH_COLLECTIVE", "host")
            warm = torch.zeros(8, device=engine.flat_g.device)
            dist.all_reduce(warm, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
    return engine


def _allreduce_any_world(flat):
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat
