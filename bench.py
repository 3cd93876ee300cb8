#!/usr/bin/env python3
"""Throughput of the NVF train step on MI355X (BASELINE.json metric: leaf-blocks/s of the NVF train step,
32^3 blocks, chanstr 8,16,8,8, ch 3).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one mini-batch decoder update (gather latents -> forward, mode 'train', q = 1 -> GT pyramid
-> 3 focal losses + rate terms -> backward -> [RCCL all-reduce of the flat decoder gradient] -> fused Adam),
NVFPCC.py:149-223 minus its logging syncs, on `--batch` blocks PER GPU (weak scaling: the global
mini-batch is batch x N).  Inputs (grids, latent table, weights) are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_MACS = {"8,16,8,8": 201190992, "16,32,16,16": 788428800}      # SURVEY.md section 2.1 / BASELINE.md section 3
CONV2_MACS = {"8,16,8,8": 134217728, "16,32,16,16": 536870912}     # per block, each of fwd / bwd-data / bwd-weight
BYTES_PER_BLOCK = {"8,16,8,8": 15050624, "16,32,16,16": 29092224}  # layer-granular HBM model, SURVEY.md 8(d)
PEAK_FP32_TFLOPS = 157.3                                           # MI355X_MICROARCH.md: fp32 vector = matrix peak
PEAK_HBM_GBS = 8000.0


def build_engine(args, device, world):
    from nvfpcc_amd import network, dist as nd
    from nvfpcc_amd.engine import TrainEngine
    from nvfpcc_amd.model import Net
    from nvfpcc_amd.seeds import synthetic_seed
    from nvfpcc_amd.synth import make_blocks
    network.reset_seed(synthetic_seed())
    net = Net(None, "Gaussian", args.ch, args.chanstr, verbose=False).to(device)
    distinct = min(args.blocks, args.distinct)
    gts, dists = make_blocks(distinct)
    reps = (args.blocks + distinct - 1) // distinct
    gt = torch.from_numpy(np.tile(gts, (reps, 1, 1, 1, 1))[:args.blocks]).float().to(device)
    dist_t = torch.from_numpy(np.tile(dists, (reps, 1, 1, 1, 1))[:args.blocks]).float().to(device)
    eng = TrainEngine(net, gt, dist_t, n_points_total=float(gt.sum().item()), lmbda=200.0, w1=10.0, w2=57.0,
                      lr=1e-3, wemb=5.0, seed=0)
    nd.attach(eng, world)
    return eng


class KernelProbe:
    """HIP-event brackets around chosen C-ABI launches (same stream).  The launch is repeated REPEAT times inside
    one bracket: with a single launch the bracket also holds the host's launch latency whenever the GPU has caught
    up with the host, which a ~0.6 ms step does."""
    REPEAT = 8

    def __init__(self):
        self.events = {}

    def wrap(self, ops, fn_name, label, match):
        orig = getattr(ops, fn_name)
        probe = self

        def wrapped(*a, **k):
            if probe.enabled and match(*a, **k):
                wb = a[0] if hasattr(a[0], "jobs") and hasattr(a[0], "offset") else None   # WgradBatch method
                state = (wb.offset, len(wb.jobs)) if wb is not None else None
                orig(*a, **k)                                   # puts the host ahead of the GPU
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(probe.REPEAT):
                    if wb is not None:                          # same slabs again: nothing accumulates
                        wb.offset = state[0]
                        del wb.jobs[state[1]:]
                    out = orig(*a, **k)
                e.record()
                probe.events.setdefault(label, []).append((s, e))
                return out
            return orig(*a, **k)
        setattr(ops, fn_name, wrapped)

    enabled = False

    def summary(self):
        return {k: float(np.mean([s.elapsed_time(e) for s, e in v])) * 1e3 / self.REPEAT
                for k, v in self.events.items()}  # us per launch


def cpu_baseline(args, seconds=15.0):
    """The oracle (CPU restatement of the reference's step) on this box's host cores: bounded sample."""
    from oracle import nvf_oracle as O
    from nvfpcc_amd.seeds import synthetic_seed
    from nvfpcc_amd.synth import make_blocks
    B = args.batch
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    gts, dists = make_blocks(B)
    gt, dist = torch.from_numpy(gts).float(), torch.from_numpy(dists).float()
    channels = tuple(int(c) for c in args.chanstr.split(","))
    tr = O.OracleTrainer(args.ch, channels, synthetic_seed(), n_leaf=B, n_points=float(gt.sum()), lr=1e-3, wemb=5.0)
    idx = torch.arange(B)
    # the step is ~1 500 small aten ops: more threads is not faster, so calibrate the thread count first
    # (2 steps each) and then time the best setting -- the baseline is the CPU at its best, not at its widest
    best, best_t = None, 1e30
    for threads in sorted({t for t in (8, 16, 32, 64, cores // 2, cores) if 1 <= t <= cores}):
        torch.set_num_threads(threads)
        tr.train_step(idx, gt, dist, q=1)   # warm-up (oneDNN primitive creation)
        t0 = time.time()
        for _ in range(2):
            tr.train_step(idx, gt, dist, q=1)
        t = (time.time() - t0) / 2
        if t < best_t:
            best, best_t = threads, t
    torch.set_num_threads(best)
    tr.train_step(idx, gt, dist, q=1)
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds or n < 3:
        tr.train_step(idx, gt, dist, q=1)
        n += 1
    dt = time.time() - t0
    return {"value": round(n * B / dt, 2), "unit": "blocks/s", "cores": best, "kind": "port",
            "sample": f"{n} train steps of batch {B} (oracle/nvf_oracle.py OracleTrainer = the reference's aten CPU ops, "
                      f"torch {torch.__version__}, best of 8/16/32/64/{cores // 2}/{cores} threads = {best}, {dt:.1f} s; "
                      f"host has {cores} logical CPUs)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--batch", type=int, default=16, help="blocks per GPU per step (reference --batchsize 16)")
    ap.add_argument("--blocks", type=int, default=917, help="leaf blocks resident per GPU (longdress l5: 917)")
    ap.add_argument("--distinct", type=int, default=128, help="distinct synthetic blocks generated, then tiled")
    ap.add_argument("--ch", type=int, default=3)
    ap.add_argument("--chanstr", default="8,16,8,8")
    ap.add_argument("--q", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="also time batch 256 and the full-batch latent step")
    ap.add_argument("--naive", action="store_true", help="debug: one-thread-per-output kernels")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the host instead of replaying a captured HIP graph")
    args = ap.parse_args()

    from nvfpcc_amd import dist as nd, ops
    rank, local_rank, world = nd.init()
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    device = torch.device("cuda", int(os.environ.get("NVF_DEVICE_OVERRIDE", local_rank)))   # override: test hook
    torch.cuda.set_device(device)
    ops.set_naive(args.naive)
    eng = build_engine(args, device, world)
    B = args.batch

    # epoch order shared by every rank (seeded), global mini-batch = B * world
    rng = np.random.default_rng(1234)
    order = np.concatenate([rng.permutation(args.blocks) for _ in range(
        (args.steps + args.warmup + 2) * B * world // args.blocks + 2)])
    counts = eng.counts

    graphed = None
    if not args.no_graph:
        from nvfpcc_amd.engine import GraphedTrainStep
        graphed = GraphedTrainStep(eng, B, args.q)

    def step(i, use_graph=True):
        ids, whole = nd.shard_minibatch(order, i, B * world, rank, world)
        if graphed is not None and use_graph:
            graphed(ids, n_pts=float(counts[whole].sum()))
        else:
            eng.train_step(ids, args.q, n_pts=float(counts[whole].sum()))

    probe = KernelProbe()
    c3 = int(args.chanstr.split(",")[3])
    # conv2 forward, backward-data (batch <= 64; VALU gather kernel above that) and weight gradient (partial-sum
    # launch of nvf_wgrad_partial) all run on the matrix cores
    probe.wrap(ops, "conv3d_k4_mfma", "conv2_fwd",
               lambda x, wp, b, pad, pair, *a, **kw: pad == 0 and x.shape[-1] == 35 and x.shape[1] == c3)
    probe.wrap(ops, "conv3d_k4_mfma", "conv2_bwd_data",
               lambda x, wp, b, pad, pair, *a, **kw: pad == 3 and x.shape[-1] == 32 and x.shape[1] == c3)
    probe.wrap(ops, "conv3d_gather", "conv2_fwd",
               lambda x, w, b, cout, k, s, p, osz, *a, **kw: k == 4 and osz[0] == 32 and x.shape[1] == c3)
    probe.wrap(ops, "conv3d_gather", "conv2_bwd_data",
               lambda x, w, b, cout, k, s, p, osz, *a, **kw: k == 4 and osz[0] == 35 and x.shape[1] == c3)
    probe.wrap(ops.WgradBatch, "add", "conv2_bwd_weight",
               lambda self_, p_, q_, k, s, pad, *a, **kw: k == 4 and s == 1 and p_.shape[-1] == 32)
    # narrow decoder: the conv2, up2 and conv1 weight gradients are ONE launch (nvf_wgrad_mfma3_partial)
    probe.wrap(ops.WgradBatch, "add_mfma3", "wgrad_conv2_up2_conv1", lambda self_, ps, qs, outs: True)
    # ... and since the five-gradient launch (nvf_wgrad_trunk5_partial) up1's and conv0's ride in it as well
    probe.wrap(ops.WgradBatch, "add_trunk5", "wgrad_trunk5", lambda self_, ps, qs, outs: True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    # kernels inside a replayed graph cannot be bracketed by events: time the dominant kernels on the same stream in
    # the same process right after the timed region, same shapes and operands, launched from the host
    probe.enabled = True
    for i in range(args.warmup + args.steps, args.warmup + args.steps + min(args.steps, 10)):
        step(i, use_graph=False)
    torch.cuda.synchronize()
    probe.enabled = False
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64,
                         device=device if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss = eng.loss_value()
    blocks_per_s = args.steps * B * world / dt

    extra = {}
    if args.sweep and rank == 0 and world == 1:
        order = np.concatenate([rng.permutation(args.blocks) for _ in range(13 * 256 // args.blocks + 2)])
        for b2 in (256,):
            for i in range(3):
                eng.train_step(order[i * b2:(i + 1) * b2], args.q)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n2 = 10
            for i in range(n2):
                eng.train_step(order[i * b2:(i + 1) * b2], args.q)
            torch.cuda.synchronize()
            extra[f"train_step_B{b2}_blocks_per_s"] = round(n2 * b2 / (time.perf_counter() - t1), 1)
        eng.latent_step(args.q)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(3):
            eng.latent_step(args.q)
        torch.cuda.synchronize()
        extra[f"latent_step_N{args.blocks}_blocks_per_s"] = round(3 * args.blocks / (time.perf_counter() - t1), 1)

    if rank == 0:
        kern_us = probe.summary()
        macs = CONV2_MACS.get(args.chanstr)
        roofline = None
        if kern_us and macs:
            single = dict(kern_us)
            label = max(single, key=single.get)      # the dominant single launch of the step
            us = single[label]
            # algorithmic MACs of what the launch computes (not the halo / padding lanes it also executes);
            # the three-gradient launch: conv2 + up2 (32 768 000 MAC/block) + conv1 (16 777 216), SURVEY 2.1
            # the five-gradient launch adds up1 (8 192 000) and conv0 (1 024 000)
            layer_macs = {"wgrad_conv2_up2_conv1": macs + 32768000 + 16777216,
                          "wgrad_trunk5": macs + 32768000 + 16777216 + 8192000 + 1024000}
            flops = 2.0 * layer_macs.get(label, macs) * B
            achieved = flops / (us * 1e-6) / 1e12
            traffic = None
            tj = os.path.join(ROOT, "profiles", "r01_traffic.json")
            if os.path.isfile(tj):     # HBM bytes per launch from the rocprofv3 --pmc passes (profiles/)
                t = json.load(open(tj))
                if t.get("batch") == B and t.get("chanstr") == args.chanstr:
                    traffic = t["hbm_bytes_per_launch"].get({"wgrad_conv2_up2_conv1": "conv2_bwd_weight",
                                                                 "wgrad_trunk5": "conv2_bwd_weight"}.get(label, label))
            roofline = {"bound": "mfma", "kernel": label, "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                        "avg_launch_us": round(us, 2), "flops_per_launch": flops,
                        "all_kernels_avg_us": {k: round(v, 2) for k, v in kern_us.items()}}
        fwd = FWD_MACS.get(args.chanstr)
        out = {
            "metric": "leaf-blocks/sec NVF train step (32^3, chanstr=%s)" % args.chanstr,
            "value": round(blocks_per_s, 1), "unit": "blocks/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "train_step: decoder mini-batch update, NVFPCC.py:149-223 (fwd mode=train q=%d, "
                                   "3 focal losses + rate terms, bwd, fused Adam)" % args.q,
                       "launch": "host" if graphed is None else "hip-graph replay + fused Adam",
                       "batch_per_gpu": B, "global_batch": B * world, "blocks_resident": args.blocks,
                       "ch": args.ch, "chanstr": args.chanstr, "parallelism": f"dp{world}",
                       "data_detail": f"{min(args.blocks, args.distinct)} distinct synthetic 32^3 quadric-sheet blocks "
                                      f"(2.5-3.5% occupancy, exact EDT distance) tiled to {args.blocks}; seed-init weights"},
            "loss_last_step": round(loss, 3),
        }
        if fwd:
            tf = blocks_per_s / world * 6.0 * fwd / 1e12
            out["step_level"] = {"fp32_tflops_per_gpu": round(tf, 3), "frac_of_fp32_peak": round(tf / PEAK_FP32_TFLOPS, 4),
                                 "hbm_fraction_layer_granular_model": round(
                                     blocks_per_s / world * BYTES_PER_BLOCK[args.chanstr] / (PEAK_HBM_GBS * 1e9), 4)}
        if roofline:
            out["roofline"] = roofline
        if extra:
            out["sweep"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
