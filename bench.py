#!/usr/bin/env python3
"""Throughput of the NVF train step on MI355X (BASELINE.json metric: leaf-blocks/s of the NVF train step,
32^3 blocks, chanstr 8,16,8,8, ch 3).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus 8                      # starts 8 ranks itself (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one mini-batch decoder update (gather latents -> forward, mode 'train', q = 1 -> GT pyramid
-> 3 focal losses + rate terms -> backward -> [RCCL all-reduce of the flat decoder gradient] -> fused Adam),
NVFPCC.py:149-223 minus its logging syncs, on `--batch` blocks PER GPU (weak scaling: the global
mini-batch is batch x N).  Inputs (grids, latent table, weights) are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  Beside the step number the line carries `epoch`: whole epochs of NVFPCC.py train
(58 mini-batches incl. the short last one + the full-batch latent step + the log line's device sums, eval on
every 10th) as end-to-end blocks/s -- what a user of the CLI sees.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_MACS = {"8,16,8,8": 201190992, "16,32,16,16": 788428800}      # SURVEY.md section 2.1 / BASELINE.md section 3
CONV2_MACS = {"8,16,8,8": 134217728, "16,32,16,16": 536870912}     # per block, each of fwd / bwd-data / bwd-weight
# per-block MACs of the other trunk layers (each of fwd / bwd-data / bwd-weight), SURVEY.md section 2.1
UP2_MACS = {"8,16,8,8": 32768000, "16,32,16,16": 131072000}
# v_mfma_f32_16x16x4_f32 instructions (2048 FLOP each) per block of the Winograd launches (SQ_INSTS_MFMA / batch,
# profiles/r04_pmc_sq_b16.md, tools/wino16_prof.py)
WINO_MFMAS = {"8,16,8,8": {"conv2_fwd": 64000, "conv2_bwd_data": 84000, "wgrad_trunk5": 158528},
              "16,32,16,16": {"conv2_fwd": 204800, "conv2_bwd_data": 268800, "conv2_bwd_weight": 204800}}
CONV1_MACS = {"8,16,8,8": 16777216, "16,32,16,16": 67108864}
UP1_MACS = {"8,16,8,8": 8192000, "16,32,16,16": 32768000}
CONV0_MACS = {"8,16,8,8": 1024000, "16,32,16,16": 4096000}
HEADS_MACS = {"8,16,8,8": 27 * (8 * 32 ** 3 + 8 * 16 ** 3 + 16 * 8 ** 3), "16,32,16,16": 27 * (16 * 32 ** 3 + 16 * 16 ** 3 + 32 * 8 ** 3)}
BYTES_PER_BLOCK = {"8,16,8,8": 15050624, "16,32,16,16": 29092224}  # layer-granular HBM model, SURVEY.md 8(d)
PEAK_FP32_TFLOPS = 157.3                                           # MI355X_MICROARCH.md: fp32 vector = matrix peak
PEAK_HBM_GBS = 8000.0


def parse_args(argv=None):
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
    ap.add_argument("--mode", choices=["step", "epoch"], default="step",
                    help="epoch: time whole NVFPCC.py-train epochs only (the default run reports both)")
    ap.add_argument("--epochs", type=int, default=10, help="epochs timed for the `epoch` object")
    ap.add_argument("--no-epoch", action="store_true", help="skip the epoch object (profiling runs: step kernels only)")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps each (value = the first)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 --pmc child passes for roofline.traffic")
    ap.add_argument("--prime-rounds", type=int, default=1, help="passes over the captured graphs before the warm-up steps")
    ap.add_argument("--no-direct", action="store_true", help="skip the `direct_form` object (the same region with TrainEngine(winograd=False))")
    ap.add_argument("--sustained-s", type=float, default=5.5, help="length of the `sustained` region in seconds (0: skip)")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip the `sweep` object (batch 256 per GPU, full-batch latent step, 4096-block latent step + eval)")
    ap.add_argument("--sweep-blocks", type=int, default=4096, help="blocks of the sweep's big latent step / eval (all ranks together)")
    ap.add_argument("--pmc-mark", default="", help=argparse.SUPPRESS)
    ap.add_argument("--naive", action="store_true", help="debug: one-thread-per-output kernels")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the host instead of replaying a captured HIP graph")
    ap.add_argument("--force-collective", action="store_true",
                    help="one rank, but with an RCCL process group and the all-reduce in the step (plumbing check)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks (one process per GPU, RCCL)
    BEFORE this process touches the GPU, and pass rank 0's JSON line through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def build_engine(args, device, world):
    import numpy as np
    import torch
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
    nd.attach(eng, world, force=args.force_collective)
    return eng


class KernelProbe:
    """HIP-event brackets around chosen C-ABI launches (same stream).  The launch is repeated REPEAT times inside
    one bracket: with a single launch the bracket also holds the host's launch latency whenever the GPU has caught
    up with the host, which a ~0.5 ms step does."""
    REPEAT = 8

    def __init__(self):
        self.events = {}

    def wrap(self, ops, fn_name, label, match):
        import torch
        from nvfpcc_amd import ops as ops_mod
        orig = getattr(ops, fn_name)
        probe = self

        def wrapped(*a, **k):
            if probe.mark and label in probe.mark.split(",") and match(*a, **k):
                # child pass under rocprofv3: a marker dispatch (uniform_kernel over MARK_N + 256 i floats, i = the
                # label's position in the list) right in front of the probed launch, so the parent finds it in the
                # counter / trace CSV by position, whatever the kernel is called
                ops_mod.uniform((probe.mark_n(probe.mark.split(",").index(label)),), a[0].device, 0, 0)
                return orig(*a, **k)
            if probe.enabled and match(*a, **k):
                wb = a[0] if hasattr(a[0], "jobs") and hasattr(a[0], "offset") else None   # WgradBatch method
                state = (wb.offset, len(wb.jobs)) if wb is not None else None
                orig(*a, **k)                                   # puts the host ahead of the GPU
                sums_done = getattr(wb, "sums_done", None)
                jobs_first = list(wb.jobs[state[1]:]) if wb is not None else None    # the step's own reduction jobs
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                k2 = k
                if wb is not None and k.get("sums") is not None:
                    # the repeats run the launch without its few dozen bias-sum workgroups (+ the two-float coefficient
                    # copy): their final pass is queued once per step, and a second partial pass would put a
                    # multi_channel_sum_final launch of its own (5 us) behind every repeat, inside the bracket
                    k2 = dict(k, sums=None, coef=None)
                for _ in range(probe.REPEAT):
                    if wb is not None:                          # same slabs again: nothing accumulates
                        wb.offset = state[0]
                        del wb.jobs[state[1]:]
                    out = orig(*a, **k2)
                e.record()
                if sums_done is not None:
                    wb.sums_done = sums_done                    # what the step's first (complete) launch left behind
                if wb is not None:
                    # ... and its job list: a repeat runs without the queued stem / up1 stages, so its conv0 job differs
                    del wb.jobs[state[1]:]
                    wb.jobs.extend(jobs_first)
                probe.events.setdefault(label, []).append((s, e))
                return out
            return orig(*a, **k)
        setattr(ops, fn_name, wrapped)

    enabled = False
    mark = None            # comma list of labels to put markers in front of (child passes)
    MARK_N = 7717          # grid of the marker dispatch of label 0: ceil(7717 / 256) workgroups of 256; label i: + i

    @classmethod
    def mark_n(cls, i):
        return cls.MARK_N + 256 * i

    def summary(self):
        import numpy as np
        return {k: float(np.mean([s.elapsed_time(e) for s, e in v])) * 1e3 / self.REPEAT
                for k, v in self.events.items()}  # us per launch


def cpu_baseline(args, seconds=18.0, samples=3):
    """The oracle (CPU restatement of the reference's step; tests/golden/trajectory.npz pins OracleTrainer to the
    reference's own loop) on this box's host cores: a bounded sample, reported as the median of `samples` timed runs
    with their spread."""
    import torch
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
    # (3 steps each) and then time the best setting -- the baseline is the CPU at its best, not at its widest
    best, best_t = None, 1e30
    for threads in sorted({t for t in (8, 16, 32, 64, cores // 2, cores) if 1 <= t <= cores}):
        torch.set_num_threads(threads)
        tr.train_step(idx, gt, dist, q=1)   # warm-up (oneDNN primitive creation)
        t0 = time.time()
        for _ in range(3):
            tr.train_step(idx, gt, dist, q=1)
        t = (time.time() - t0) / 3
        if t < best_t:
            best, best_t = threads, t
    torch.set_num_threads(best)
    tr.train_step(idx, gt, dist, q=1)
    rates, total_n, total_t = [], 0, 0.0
    for _ in range(samples):
        t0 = time.time()
        n = 0
        while time.time() - t0 < seconds / samples or n < 3:
            tr.train_step(idx, gt, dist, q=1)
            n += 1
        dt = time.time() - t0
        rates.append(n * B / dt)
        total_n, total_t = total_n + n, total_t + dt
    return {"value": round(statistics.median(rates), 2), "unit": "blocks/s", "cores": best, "kind": "port",
            "min": round(min(rates), 2), "max": round(max(rates), 2), "samples": [round(r, 2) for r in rates],
            "sample": f"median of {samples} runs, {total_n} train steps of batch {B} in all (oracle/nvf_oracle.py "
                      f"OracleTrainer = the reference's aten CPU ops, pinned to the reference's own loop by "
                      f"tests/golden/trajectory.npz; torch {torch.__version__}, best of 8/16/32/64/{cores // 2}/{cores} "
                      f"threads = {best}, {total_t:.1f} s; host has {cores} logical CPUs)"}


def kernel_source_hash():
    """Identifies the kernel sources a PMC measurement belongs to (stale profiles are refused)."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "nvfpcc_amd", "csrc", "*.h*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measure_traffic(args, label):
    """HBM bytes per launch of the dominant kernel, measured NOW: two child runs of this script (host-launched steps)
    under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (the two do not fit one pass), medians per dispatch,
    FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM section).  The child puts a
    marker dispatch (KernelProbe.mark) in front of every launch behind `label`; the launch is the row after the marker
    in dispatch order, so nothing here depends on how the kernel or its template arguments are spelled.
    Returns (bytes or None, note, kernel name or None)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(rocprof):
        return None, "rocprofv3 not found", None
    vals, name = {}, None
    base = tempfile.mkdtemp(prefix="nvf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    mark_grid = (KernelProbe.MARK_N + 255) // 256 * 256
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(base, counter)
        cmd = [rocprof, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--",
               "python3", os.path.abspath(__file__), "--pmc-child", "--pmc-mark", label, "--no-cpu-baseline", "--no-pmc",
               "--no-graph", "--no-sweep", "--no-epoch", "--steps", "6", "--warmup", "3", "--repeats", "1", "--batch",
               str(args.batch), "--blocks", str(args.blocks), "--distinct", str(args.distinct), "--ch", str(args.ch),
               "--chanstr", args.chanstr, "--q", str(args.q)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                               timeout=240)
        except subprocess.TimeoutExpired:
            shutil.rmtree(base, ignore_errors=True)
            return None, f"rocprofv3 --pmc {counter} child timed out", None
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            shutil.rmtree(base, ignore_errors=True)
            return None, f"rocprofv3 --pmc {counter} child failed (rc {r.returncode}): {r.stdout[-300:]}", None
        rows = [row for row in csv.DictReader(open(files[0])) if row["Counter_Name"] == counter]
        rows.sort(key=lambda row: int(row["Dispatch_Id"]))
        per = []
        for prev, row in zip(rows, rows[1:]):
            if "uniform_kernel" in prev["Kernel_Name"] and int(prev["Grid_Size"]) in (mark_grid, mark_grid // 256):
                per.append(float(row["Counter_Value"]))
                name = row["Kernel_Name"]
        if not per:
            shutil.rmtree(base, ignore_errors=True)
            return None, f"no marked dispatch of {label} in the {counter} pass", None
        vals[counter] = statistics.median(per)
    shutil.rmtree(base, ignore_errors=True)
    # rocprofv3 reports both in KB
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, (
        f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command (host-launched steps), median per "
        f"dispatch of the launch behind a marker dispatch, 2 x FETCH_SIZE + WRITE_SIZE; kernel sources "
        f"{kernel_source_hash()}"), name


def measure_in_step_us(args, labels):
    """Duration of the launches behind `labels` AS THE STEP ISSUES THEM (the five-gradient launch with the stem's backward,
    its bias-sum workgroups, coefficient copy and the queued latent tail): one child run of this script -- the step graphs
    replayed as in the timed region, a marker dispatch captured in front of each probed launch (host-launched steps with
    --no-graph) -- under `rocprofv3 --kernel-trace` (no counters: kernels are not serialised
    beyond the stream's own order); per label the median End - Start of the dispatch that follows its marker.
    Returns ({label: us}, {label: kernel name}, note)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(rocprof):
        return {}, {}, "rocprofv3 not found"
    base = tempfile.mkdtemp(prefix="nvf_trace_", dir="/tmp")
    cmd = [rocprof, "--kernel-trace", "-d", base, "-o", "kt", "--output-format", "csv", "--",
           "python3", os.path.abspath(__file__), "--pmc-child", "--pmc-mark", ",".join(labels), "--no-cpu-baseline",
           "--no-pmc", "--no-sweep", "--no-epoch", "--steps", "48", "--warmup", "4", "--repeats", "1",
           "--batch", str(args.batch), "--blocks", str(args.blocks), "--distinct", str(args.distinct), "--ch", str(args.ch),
           "--chanstr", args.chanstr, "--q", str(args.q)] + (["--no-graph"] if args.no_graph else [])
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        shutil.rmtree(base, ignore_errors=True)
        return {}, {}, "rocprofv3 --kernel-trace child timed out"
    files = glob.glob(os.path.join(base, "**", "*kernel_trace.csv"), recursive=True)
    if r.returncode != 0 or not files:
        shutil.rmtree(base, ignore_errors=True)
        return {}, {}, f"rocprofv3 --kernel-trace child failed (rc {r.returncode}): {r.stdout[-300:]}"
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda row: int(row["Start_Timestamp"]))
    grids = {(KernelProbe.mark_n(i) + 255) // 256 * 256: lab for i, lab in enumerate(labels)}
    per, names = {lab: [] for lab in labels}, {}
    for prev, row in zip(rows, rows[1:]):
        if "uniform_kernel" in prev["Kernel_Name"]:
            g = int(prev.get("Grid_Size") or prev.get("Grid_Size_X") or 0)       # threads (kernel trace: Grid_Size_X)
            lab = grids.get(g) or grids.get(g * 256)
            if lab is not None:
                per[lab].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000.0)
                names[lab] = row["Kernel_Name"]
    shutil.rmtree(base, ignore_errors=True)
    return ({lab: statistics.median(v) for lab, v in per.items() if v}, names,
            "in-step: rocprofv3 --kernel-trace child pass of this command (%s), median duration of the dispatch behind a "
            "marker dispatch" % ("host-launched steps" if args.no_graph else
                                 "the same unrolled step graphs replayed, marker dispatches captured into them"))


def run(args):
    # stdout carries ONE line, the JSON: libraries that print there (RCCL writes a version banner to stdout when its
    # communicator is built) go to stderr for the life of the process; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import numpy as np
    import torch
    from nvfpcc_amd import dist as nd, ops
    from nvfpcc_amd.engine import GraphedTrainStep, EpochDriver
    if args.force_collective and "WORLD_SIZE" not in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
    rank, local_rank, world = nd.init()
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    device = torch.device("cuda", int(os.environ.get("NVF_DEVICE_OVERRIDE", local_rank)))   # override: test hook
    torch.cuda.set_device(device)
    if args.force_collective and not torch.distributed.is_initialized():
        torch.distributed.init_process_group(backend=os.environ.get("NVF_DIST_BACKEND", "nccl"), rank=0, world_size=1)
    if torch.distributed.is_initialized():
        world = torch.distributed.get_world_size()          # the live process group, not the command line
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but the process group has {world} rank(s)", file=sys.stderr)
    ops.set_naive(args.naive)
    eng = build_engine(args, device, world)
    B = args.batch
    nreg = max(args.repeats, 1)

    # epoch order shared by every rank (seeded), global mini-batch = B * world
    rng = np.random.default_rng(1234)
    order = np.concatenate([rng.permutation(args.blocks) for _ in range(
        (args.steps * nreg + args.warmup + 12) * B * world // args.blocks + 2)])
    counts = eng.counts

    probe = KernelProbe()
    c3 = int(args.chanstr.split(",")[3])
    # conv2 forward, backward-data (batch <= 64; VALU gather kernel above that) and weight gradient (partial-sum
    # launch of nvf_wgrad_partial) all run on the matrix cores
    probe.wrap(ops, "conv3d_k4_mfma", "conv2_fwd",
               lambda x, wp, b, pad, pair, *a, **kw: pad == 0 and x.shape[-1] == 35 and x.shape[1] == c3)
    probe.wrap(ops, "conv3d_k4_mfma", "conv2_bwd_data",
               lambda x, wp, b, pad, pair, *a, **kw: pad == 3 and x.shape[-1] == 32 and x.shape[1] == c3)
    # ... in the Winograd (y, x) form since round 4 (conv_wino.hip): training-step forward and backward-data
    probe.wrap(ops, "conv3d_k4_wino_fwd", "conv2_fwd", lambda x, wp, b, *a, **kw: x.shape[-1] == 35)
    probe.wrap(ops, "conv3d_k4_wino_bwd", "conv2_bwd_data", lambda dy, wp, m, *a, **kw: dy.shape[-1] == 32)
    probe.wrap(ops, "conv3d_gather", "conv2_fwd",
               lambda x, w, b, cout, k, s, p, osz, *a, **kw: k == 4 and osz[0] == 32 and x.shape[1] == c3)
    probe.wrap(ops, "conv3d_gather", "conv2_bwd_data",
               lambda x, w, b, cout, k, s, p, osz, *a, **kw: k == 4 and osz[0] == 35 and x.shape[1] == c3)
    # wide decoder: 16 output channels are the MFMA rows (conv16_mfma.hip)
    probe.wrap(ops, "conv3d_g16_mfma", "conv2_fwd",
               lambda x, wp, b, cout, k, s, pad, osz, *a, **kw: k == 4 and pad == 0 and x.shape[-1] == 35)
    probe.wrap(ops, "conv3d_g16_mfma", "conv2_bwd_data",
               lambda x, wp, b, cout, k, s, pad, osz, *a, **kw: k == 4 and pad == 3 and x.shape[-1] == 32)
    # ... in the Winograd (y, x) form since round 4 (conv16_wino.hip, wgrad16_wino.hip)
    probe.wrap(ops, "conv3d_k4_wino16_fwd", "conv2_fwd", lambda x, wp, b, *a, **kw: x.shape[-1] == 35)
    probe.wrap(ops, "conv3d_k4_wino16_bwd", "conv2_bwd_data", lambda dy, wp, m, *a, **kw: dy.shape[-1] == 32)
    probe.wrap(ops, "wgrad16_k4_wino_partial", "conv2_bwd_weight", lambda dy, x, slabs, *a, **kw: dy.shape[-1] == 32)
    probe.wrap(ops.WgradBatch, "add", "conv2_bwd_weight",
               lambda self_, p_, q_, k, s, pad, *a, **kw: k == 4 and s == 1 and p_.shape[-1] == 32)
    # narrow decoder: the conv2, up2 and conv1 weight gradients are ONE launch (nvf_wgrad_mfma3_partial)
    probe.wrap(ops.WgradBatch, "add_mfma3", "wgrad_conv2_up2_conv1", lambda self_, ps, qs, outs: True)
    # ... and since the five-gradient launch (nvf_wgrad_trunk5_partial) up1's and conv0's ride in it as well
    probe.wrap(ops.WgradBatch, "add_trunk5", "wgrad_trunk5", lambda self_, ps, qs, outs, **kw: True)
    if args.pmc_child and args.pmc_mark and not args.no_graph:
        # kernel-trace child pass: the marker dispatches are CAPTURED into the step graphs, so the probed launches are timed
        # exactly where the headline runs them -- inside replayed graphs
        probe.mark = args.pmc_mark
    graphed = None
    primed_steps = 0
    if not args.no_graph and args.mode == "step":
        # Steps per replayed graph: the engine's 16 / 8 / 4 / 2 -- and K itself when the timed region is short (K <= 64) and
        # not a multiple of 16: switching from one graph to a DIFFERENT one costs ~40 us of idle GPU on this stack
        # (tools/region_overhead.py: a 4-step + a 16-step graph = 20 steps + 70 us, one graph + 24 us), which a 20-step
        # region would carry as 2 us per step; NVFPCC.py train replays the same 16-step graph 57 times per epoch.
        unroll = None
        if args.steps <= 64 and args.steps % 16 != 0 and args.steps > 1:
            unroll = tuple(sorted(set(GraphedTrainStep.UNROLL) | {args.steps}, reverse=True))
        graphed = GraphedTrainStep(eng, B, args.q, unroll=unroll)
        # setup, like the capture itself: every graph launched before the W warm-up steps (real optimiser steps, reported
        # as config.primed_steps).  --prime-rounds > 1 keeps the GPU busy for that many passes over the graphs first.
        primed_steps = graphed.prime(args.prime_rounds)

    def shares(i):
        ids, whole = nd.shard_minibatch(order, i, B * world, rank, world)
        return ids, float(counts[whole].sum())

    def step(i, use_graph=True):
        ids, n_pts = shares(i)
        if graphed is not None and use_graph:
            graphed(ids, n_pts=n_pts)
        else:
            eng.train_step(ids, args.q, n_pts=n_pts)

    def stage_steps(lo, hi):
        """Host half of the schedule of steps [lo, hi) (which blocks, n_pts, noise steps, Adam coefficients as rows in
        pinned memory): prepared BEFORE the timed region, as a training loop prepares the next epoch's schedule while the
        GPU runs this one -- it is input data.  Returns None when the steps do not fit one upload or there is no graph."""
        if graphed is None or hi - lo > graphed.CAP or hi <= lo:
            return None
        W = B * world
        whole = np.asarray(order[lo * W:hi * W], np.int64).reshape(hi - lo, W)
        return graphed.stage_schedule((whole[:, rank::world], counts[whole].sum(axis=1).astype(np.float64)))

    def run_steps(lo, hi, staged=None):
        """Steps [lo, hi): with the graph, ONE host-to-device copy of their schedule ((hi - lo) x (B + 3) words; the copy is
        inside the timed region, the rows were filled before it when `staged` is given), then the unrolled graph replays;
        without the graph, host-launched steps."""
        if graphed is None:
            for i in range(lo, hi):
                step(i)
            return
        if staged is not None:
            graphed.load_schedule(staged)
            graphed.replay_all()
            return
        i = lo
        W = B * world
        while i < hi:
            e = min(hi, i + graphed.CAP)
            whole = np.asarray(order[i * W:e * W], np.int64).reshape(e - i, W)        # the steps' mini-batches
            graphed.load_schedule((whole[:, rank::world], counts[whole].sum(axis=1).astype(np.float64)))
            graphed.replay_all()
            i = e

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed_max(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64,
                             device=device if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    region_ms = []
    dt = None
    if args.mode == "step":
        run_steps(0, args.warmup)
        nxt = args.warmup
        for r in range(nreg):                     # region 0 is the contract's measurement; the others show its spread
            staged = stage_steps(nxt, nxt + args.steps)
            barrier()
            t0 = time.perf_counter()
            run_steps(nxt, nxt + args.steps, staged)
            t_host = time.perf_counter() - t0
            barrier()
            d = timed_max(time.perf_counter() - t0)
            if os.environ.get("NVF_BENCH_DEBUG"):
                print(f"[bench] region {r}: host enqueue {t_host * 1e6:.0f} us, total {d * 1e6:.0f} us", file=sys.stderr)
            nxt += args.steps
            region_ms.append(d / args.steps * 1e3)
            if dt is None:
                dt = d
        # kernels inside a replayed graph cannot be bracketed by events: time the dominant kernels on the same stream
        # in the same process right after the timed region, same shapes and operands, launched from the host
        if not (args.pmc_child and graphed is not None):        # (a graph-mode child pass has its markers in the graphs)
            probe.enabled = not args.pmc_child
            probe.mark = args.pmc_mark or None
            for i in range(nxt, nxt + min(args.steps, 10)):
                step(i, use_graph=False)
            torch.cuda.synchronize()
            probe.enabled = False
        loss = eng.loss_value()
    if args.pmc_child:
        return 0

    # ---- `sustained`: ONE region of >= --sustained-s seconds of the same steps through the same unrolled graphs (the
    # headline region is a few milliseconds: the reference trains 501 epochs, NVFPCC.py:128).  Ten equal parts separated
    # by HIP events on the compute stream (no host synchronisation inside the region); ms / step of the first and the
    # last tenth show whether the clock holds.
    sustained_obj = None
    if args.mode == "step" and graphed is not None and args.sustained_s > 0 and region_ms:
        per_part = max(int(args.sustained_s / 10 / (region_ms[0] * 1e-3)) // 16 * 16, 16)
        per_part = min(per_part, graphed.CAP // 16 * 16)
        srng = np.random.default_rng(77)
        W = B * world

        def part_schedule():
            need = per_part * W
            o = np.concatenate([srng.permutation(args.blocks) for _ in range(need // args.blocks + 1)])[:need]
            whole = o.reshape(per_part, W).astype(np.int64)
            return graphed.stage_schedule((whole[:, rank::world], counts[whole].sum(axis=1).astype(np.float64)))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        barrier()
        t0 = time.perf_counter()
        evs[0].record()
        for part in range(10):
            graphed.load_schedule(part_schedule())       # (a handle is staged while the previous part still runs)
            graphed.replay_all()
            evs[part + 1].record()
        barrier()
        wall = timed_max(time.perf_counter() - t0)
        parts_ms = [evs[i].elapsed_time(evs[i + 1]) / per_part for i in range(10)]
        nsteps = 10 * per_part
        sustained_obj = {"seconds": round(wall, 3), "steps": nsteps, "ms_per_step": round(wall / nsteps * 1e3, 4),
                         "blocks_per_s": round(nsteps * W / wall, 1),
                         "ms_per_step_first_tenth": round(parts_ms[0], 4), "ms_per_step_last_tenth": round(parts_ms[-1], 4),
                         "ms_per_step_tenths": [round(x, 4) for x in parts_ms],
                         "loss_last_step": round(eng.loss_value(), 3),
                         "parameters_finite": bool(torch.isfinite(eng.flat_p).all().item()),
                         "note": "one timed region, wall clock between two barriers (max over ranks); the tenths are HIP-event "
                                 "intervals on the compute stream of this rank; %d-step schedules uploaded between the parts "
                                 "without synchronising" % per_part}

    # ---- `direct_form`: the same K-step region with TrainEngine(winograd=False) -- every 4^3 layer of the training step in the
    # direct summation order, the form that follows the reference's own three-epoch trajectory to 2e-6
    # (tests/test_gpu_engine.py::test_direct_forms_follow_the_reference_trajectory_to_rounding); the headline runs the
    # reduced-multiplication (Winograd) forms, whose trajectory parity is statistical
    direct_obj = None
    if args.mode == "step" and graphed is not None and not args.no_direct:
        from nvfpcc_amd import engine as _E
        saved = _E._WINO
        _E._WINO = False
        try:
            eng_d = build_engine(args, device, world)
        finally:
            _E._WINO = saved
        assert not eng_d.winograd
        g_d = GraphedTrainStep(eng_d, B, args.q)
        g_d.prime()
        W = B * world
        drng = np.random.default_rng(55)

        def direct_region(n):
            o = np.concatenate([drng.permutation(args.blocks) for _ in range(n * W // args.blocks + 1)])[:n * W]
            whole = o.reshape(n, W).astype(np.int64)
            h = g_d.stage_schedule((whole[:, rank::world], counts[whole].sum(axis=1).astype(np.float64)))
            barrier()
            t1 = time.perf_counter()
            g_d.load_schedule(h)
            g_d.replay_all()
            barrier()
            return timed_max(time.perf_counter() - t1) / n
        direct_region(args.warmup if args.warmup > 0 else 1)
        d_ms = [direct_region(args.steps) * 1e3 for _ in range(nreg)]
        direct_obj = {"ms_per_step": round(d_ms[0], 4), "blocks_per_s": round(W / (d_ms[0] * 1e-3), 1),
                      "repeats_ms_per_step": [round(x, 4) for x in d_ms], "steps": args.steps,
                      "engine": "TrainEngine(winograd=False): conv2 / conv1 forward, backward-data and weight gradients in the "
                                "direct fixed summation order (rounds 1-3 kernels); same launch path as the headline",
                      "parity": "reference trajectory golden to 2e-6 (strict); the headline engine: statistical (5 seeds)"}
        del eng_d, g_d
        torch.cuda.empty_cache()

    # ---- whole epochs of NVFPCC.py train: mini-batches (graph replay; the short last batch replays a graph of its own size), the
    # full-batch latent step on this rank's shard, the device-side sums of the log line, eval on every 10th epoch
    epoch_obj = None
    if (args.mode == "epoch" or not args.no_graph) and not (args.no_epoch and args.mode == "step"):
        N = args.blocks
        drv = EpochDriver(eng, B * world, rank, world, use_graph=not args.no_graph)
        lo, hi = nd.shard_range(N, rank, world)
        erng = np.random.default_rng(99)

        def one_epoch(ep):
            n = drv.run(erng.permutation(N), args.q)
            if hi > lo:
                eng.latent_step(args.q, lo, hi)
            else:
                eng.noise_step += 1
            nd.allgather_rows_(eng.emb, rank, world)
            eng.read_epoch_stats(reduce=nd.allreduce_sum_ if world > 1 else None, world=world)   # the epoch's one sync
            if ep % 10 == 0 and rank == 0:
                a = eng.eval_forward(q=2)
                ops.metrics(a["p2"], eng.gt, eng.dist, 0.5, 0.6).cpu()
            return n
        one_epoch(1)
        one_epoch(2)
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):                         # the full-batch latent step alone (this rank's shard)
            if hi > lo:
                eng.latent_step(args.q, lo, hi)
        barrier()
        latent_ms = timed_max(time.perf_counter() - t0) / 3 * 1e3
        t0 = time.perf_counter()
        nst = 0
        for ep in range(args.epochs):
            nst = one_epoch(ep)
        barrier()
        de = timed_max(time.perf_counter() - t0)
        step_ms = region_ms[0] if region_ms else None
        epoch_obj = {"epochs": args.epochs, "ms_per_epoch": round(de / args.epochs * 1e3, 3),
                     "blocks_per_s": round(N * args.epochs / de, 1), "minibatches_per_epoch": nst,
                     "includes": "mini-batch steps (unrolled graph replays; the short last batch replays a single-step graph of its own size), full-batch latent "
                                 "step, per-epoch stats read-back, eval forward + metrics on every 10th epoch",
                     "latent_step_ms": round(latent_ms, 3),
                     "bound_steps_plus_latent_plus_10pct_ms": None if step_ms is None else round(
                         1.1 * (nst * step_ms + latent_ms), 3)}
        if args.mode == "epoch":
            dt = de
            loss = float("nan")

    blocks_per_s = (args.steps * B * world / dt) if args.mode == "step" else args.blocks * args.epochs / dt

    # ---- sweep (B in {16, 256, 4096}; 8(e): report both the latency-bound B = 16 line and the lines
    # that keep every GPU busy): batch 256 PER GPU train step, the full-batch latent step over this rank's shard of
    # the resident blocks, and a `--sweep-blocks`-block latent step + eval forward sharded over the ranks.  Weak scaling
    # for the first, strong for the other two; every figure is a max over ranks and carries its step-level fraction.
    launch_desc = "host" if (graphed is None and args.mode == "step") else (
        "hip-graph replay (step head .. backward), then all-reduce + Adam from the host"
        if (graphed is not None and graphed.collective == "host") else
        "hip-graph replay (step head .. Adam in one graph, %s steps per replay; per-step scalars from a device-resident "
        "schedule whose rows are filled in pinned host memory before the timed region and copied to the device inside it)" % ("/".join(str(u) for u in graphed.unrolls if u in graphed.graphs_u) if graphed is not None and graphed.graphs_u else "1"))
    collective_desc = None if eng.grad_hook is None else (
        "all-reduce captured in the step graph" if (graphed is not None and graphed.collective == "graph")
        else "all-reduce launched from the host")
    extra = {}
    if not args.no_sweep and args.mode == "step":
        fwd_macs = FWD_MACS.get(args.chanstr, 0)
        frac = lambda bps, flop_per_block: round(bps / world * flop_per_block / 1e12 / PEAK_FP32_TFLOPS, 4)

        def timed(fn, n):
            fn()
            barrier()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            barrier()
            return timed_max(time.perf_counter() - t1) / n

        # strong scaling of the reference's own optimisation problem: the GLOBAL mini-batch stays at --batch (16) and is split
        # over the ranks (position j -> rank j mod W); at N = 1 this is the headline itself
        if world > 1 and graphed is not None and B % world == 0:
            share = B // world
            g_s = GraphedTrainStep(eng, share, args.q)
            if g_s.graphs_u:
                g_s.prime()
            nst = max(min(args.steps, 64), 4)
            o_s = np.concatenate([rng.permutation(args.blocks) for _ in range(nst * B // args.blocks + 2)])[:nst * B]
            whole_s = o_s.reshape(nst, B).astype(np.int64)
            npts_s = counts[whole_s].sum(axis=1).astype(np.float64)

            def strong_steps():
                g_s.load_schedule((whole_s[:, rank::world], npts_s))
                g_s.replay_all()
            dts = timed(strong_steps, 3) / nst
            extra[f"train_step_B{B}_global_strong"] = {
                "blocks_per_s": round(B / dts, 1), "ms_per_step": round(dts * 1e3, 4), "global_batch": B,
                "blocks_per_gpu": share, "scaling": "strong (the reference's batch of %d split over the ranks, one all-reduce "
                                                    "of the decoder gradients per step)" % B,
                "frac_of_fp32_peak": frac(B / dts, 6.0 * fwd_macs)}
            del g_s
        b2 = 256
        order2 = np.concatenate([rng.permutation(args.blocks) for _ in range(14 * b2 * world // args.blocks + 2)])
        it = [0]

        def big_step():
            ids, whole = nd.shard_minibatch(order2, it[0], b2 * world, rank, world)
            it[0] += 1
            eng.train_step(ids, args.q, n_pts=float(counts[whole].sum()))
        for _ in range(2):
            big_step()
        dt2 = timed(big_step, 8)
        bps = b2 * world / dt2
        extra[f"train_step_B{b2}_per_gpu"] = {"blocks_per_s": round(bps, 1), "ms_per_step": round(dt2 * 1e3, 3),
                                              "global_batch": b2 * world, "scaling": "weak",
                                              "frac_of_fp32_peak": frac(bps, 6.0 * fwd_macs)}
        lo, hi = nd.shard_range(args.blocks, rank, world)
        dt3 = timed(lambda: eng.latent_step(args.q, lo, hi) if hi > lo else None, 3)
        bps = args.blocks / dt3
        extra[f"latent_step_N{args.blocks}"] = {"blocks_per_s": round(bps, 1), "ms_per_step": round(dt3 * 1e3, 3),
                                                 "scaling": "strong (blocks sharded over the ranks, no collective)",
                                                 "frac_of_fp32_peak": frac(bps, 4.0 * fwd_macs)}
        nbig = args.sweep_blocks
        if nbig > 0:
            del eng, graphed
            torch.cuda.empty_cache()
            lo, hi = nd.shard_range(nbig, rank, world)
            big = argparse.Namespace(**dict(vars(args), blocks=max(hi - lo, 1), force_collective=False))
            eng2 = build_engine(big, device, 1)
            dt4 = timed(lambda: eng2.latent_step(args.q), 3)
            dt5 = timed(lambda: eng2.eval_forward(), 3)
            extra[f"latent_step_N{nbig}"] = {"blocks_per_s": round(nbig / dt4, 1), "ms_per_step": round(dt4 * 1e3, 2),
                                             "blocks_per_gpu": hi - lo, "scaling": "strong",
                                             "frac_of_fp32_peak": frac(nbig / dt4, 4.0 * fwd_macs)}
            extra[f"eval_forward_N{nbig}"] = {"blocks_per_s": round(nbig / dt5, 1), "ms": round(dt5 * 1e3, 2),
                                              "blocks_per_gpu": hi - lo, "scaling": "strong",
                                              "frac_of_fp32_peak": frac(nbig / dt5, 2.0 * fwd_macs),
                                              "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
            del eng2
            # BASELINE configs[4]'s decoder (--ch 8 --chanstr 16,32,16,16 --wemb 8) at the reference's batch of 16:
            # the same train step through the same launch path as the headline (unrolled graphs, device-resident schedule)
            if args.chanstr == "8,16,8,8" and world == 1 and not args.no_graph:
                torch.cuda.empty_cache()
                wide = argparse.Namespace(**dict(vars(args), ch=8, chanstr="16,32,16,16", force_collective=False))
                eng3 = build_engine(wide, device, 1)
                g3 = GraphedTrainStep(eng3, B, args.q)
                g3.prime()
                nst = 20
                whole3 = np.concatenate([rng.permutation(args.blocks) for _ in range(nst * B // args.blocks + 2)])[:nst * B]
                whole3 = whole3.reshape(nst, B).astype(np.int64)
                npts3 = counts[whole3].sum(axis=1).astype(np.float64)

                def wide_steps():
                    g3.load_schedule((whole3, npts3))
                    g3.replay_all()
                dt6 = timed(wide_steps, 3) / nst
                wm = FWD_MACS["16,32,16,16"]
                extra["train_step_wide_B16"] = {"blocks_per_s": round(B / dt6, 1), "ms_per_step": round(dt6 * 1e3, 4),
                                                "ch": 8, "chanstr": "16,32,16,16", "steps": nst,
                                                "frac_of_fp32_peak": round(B / dt6 * 6.0 * wm / 1e12 / PEAK_FP32_TFLOPS, 4)}
                del eng3, g3

    if rank == 0:
        kern_us = probe.summary()
        macs = CONV2_MACS.get(args.chanstr)
        roofline = None
        if kern_us and macs:
            single = dict(kern_us)
            label = max(single, key=single.get)      # the dominant single launch of the step
            us = single[label]
            # algorithmic MACs of what the launch computes (not the halo / padding lanes it also executes);
            # the three-gradient launch: conv2 + up2 + conv1; the five-gradient launch adds up1 and conv0
            cs = args.chanstr
            from nvfpcc_amd import engine as _E
            heads_in = _E._HEADS_IN_TRUNK5 and cs == "8,16,8,8"     # the three heads' gradients ride in that launch
            layer_macs = {"wgrad_conv2_up2_conv1": macs + UP2_MACS[cs] + CONV1_MACS[cs],
                          "wgrad_trunk5": macs + UP2_MACS[cs] + CONV1_MACS[cs] + UP1_MACS[cs] + CONV0_MACS[cs]
                                          + (HEADS_MACS[cs] if heads_in else 0)}
            flops = 2.0 * layer_macs.get(label, macs) * B
            # the launch exactly as the step issues it (bias-sum workgroups, coefficient copy, queued latent tail): its
            # duration inside host-launched steps from a kernel-trace child pass; the HIP-event bracket around 8 repeats
            # (a slightly lighter launch: no bias sums, no tail) stays in the line as `event_bracket_us`
            labels = [label] + [l for l in ("conv2_fwd", "conv2_bwd_data") if l in single and l != label]
            in_step, in_names, in_note = ({}, {}, "not measured (--no-pmc or N > 1)")
            if world == 1 and not args.no_pmc:
                torch.cuda.synchronize()
                in_step, in_names, in_note = measure_in_step_us(args, labels)
            us_step = in_step.get(label, us)
            achieved = flops / (us_step * 1e-6) / 1e12
            traffic, tnote, kname = None, "not measured (--no-pmc or N > 1)", None
            if world == 1 and not args.no_pmc:
                torch.cuda.synchronize()
                traffic, tnote, kname = measure_traffic(args, label)
            per_kernel = {}
            # (the five-gradient launch's count was measured at batch 16: its workgroup caps make it non-linear in the batch)
            wino = {k: v for k, v in WINO_MFMAS.get(cs, {}).items() if k != "wgrad_trunk5" or B == 16} if _E._WINO else {}
            for lab in labels:
                f = flops if lab == label else 2.0 * macs * B
                u = in_step.get(lab, single[lab])
                ent = {"us": round(u, 2), "flops": f, "tflops": round(f / (u * 1e-6) / 1e12, 2),
                       "frac": round(f / (u * 1e-6) / 1e12 / PEAK_FP32_TFLOPS, 4),
                       "source": "in-step kernel trace" if lab in in_step else "HIP-event bracket, 8 repeats",
                       "kernel_name": in_names.get(lab)}
                if lab in wino:
                    # a reduced-multiplication (Winograd) launch: `flops` / `frac` are the layer's direct-form (algorithmic)
                    # count, which the kernel does not execute -- and which can therefore exceed the pipe's peak; the
                    # fraction of the pipe it really occupies is the MFMA work it issues
                    ex = wino[lab] * B * 2048.0
                    ent.update({"form": "Winograd (y, x): 25 products per 2 x 2 outputs and z tap instead of 64",
                                "mfma_flops_executed": ex,
                                "executed_frac": round(ex / (u * 1e-6) / 1e12 / PEAK_FP32_TFLOPS, 4)})
                per_kernel[lab] = ent
            roofline = {"bound": "mfma", "kernel": label, "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                        "traffic": None if traffic is None else round(traffic), "traffic_source": tnote,
                        "kernel_name": kname or in_names.get(label),
                        "avg_launch_us": round(us_step, 2), "avg_launch_us_source": in_note,
                        "event_bracket_us": round(us, 2), "flops_per_launch": flops,
                        "all_kernels": per_kernel,
                        **({"mfma_flops_executed": wino[label] * B * 2048.0,
                            "executed_frac": round(wino[label] * B * 2048.0 / (us_step * 1e-6) / 1e12 / PEAK_FP32_TFLOPS, 4),
                            "executed_note": "frac = the launch's algorithmic (direct-form) FLOPs over its time; part of it runs "
                                             "in a Winograd form, so the MFMA work it executes is less (SQ_INSTS_MFMA x 2048)"}
                           if label in wino else {}),
                        "all_kernels_avg_us": {k: round(v, 2) for k, v in kern_us.items()},
                        "note": "fp32 VALU instructions do not hide behind fp32 MFMAs on gfx950 (profiles/"
                                "r04_mfma_valu_overlap.md): a kernel's time is 32 cycles per MFMA plus 2.5-5 per vector "
                                "instruction, in series"}
        fwd = FWD_MACS.get(args.chanstr)
        workload = ("train_step: decoder mini-batch update, NVFPCC.py:149-223 (fwd mode=train q=%d, 3 focal losses + "
                    "rate terms, bwd, fused Adam)" % args.q) if args.mode == "step" else (
                   "train epoch: NVFPCC.py:128-292 (mini-batch updates, full-batch latent step, eval every 10th)")
        out = {
            "metric": "leaf-blocks/sec NVF train step (32^3, chanstr=%s)" % args.chanstr,
            "value": round(blocks_per_s, 1), "unit": "blocks/s", "n_gpus": world,
            "steps": args.steps if args.mode == "step" else args.epochs,
            "warmup": args.warmup, "ms_per_step": round(dt / (args.steps if args.mode == "step" else args.epochs) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak" if args.mode == "step" else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "launch": launch_desc, "collective": collective_desc,
                       "primed_steps": primed_steps,      # optimiser steps run before the W warm-up steps (every captured
                       # graph launched once): parameters / Adam moments / noise counters have advanced by that many
                       "batch_per_gpu": B, "global_batch": B * world, "blocks_resident": args.blocks,
                       "ch": args.ch, "chanstr": args.chanstr, "parallelism": f"dp{world}",
                       "data_detail": f"{min(args.blocks, args.distinct)} distinct synthetic 32^3 quadric-sheet blocks "
                                      f"(2.5-3.5% occupancy, exact EDT distance) tiled to {args.blocks}; seed-init weights"},
        }
        if args.mode == "step":
            out["loss_last_step"] = round(loss, 3)
            out["repeats"] = {"ms_per_step": [round(x, 4) for x in region_ms], "min": round(min(region_ms), 4),
                              "median": round(statistics.median(region_ms), 4),
                              "note": "value / ms_per_step come from the first region (the contract's K steps)"}
        if fwd and args.mode == "step":
            tf = blocks_per_s / world * 6.0 * fwd / 1e12
            out["step_level"] = {"fp32_tflops_per_gpu": round(tf, 3), "frac_of_fp32_peak": round(tf / PEAK_FP32_TFLOPS, 4),
                                 "hbm_fraction_layer_granular_model": round(
                                     blocks_per_s / world * BYTES_PER_BLOCK[args.chanstr] / (PEAK_HBM_GBS * 1e9), 4)}
        if roofline:
            out["roofline"] = roofline
        if direct_obj:
            out["direct_form"] = direct_obj
        if sustained_obj:
            out["sustained"] = sustained_obj
        if epoch_obj:
            out["epoch"] = epoch_obj
        if extra:
            out["sweep"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), file=json_out, flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))          # nothing above has touched the GPU
    sys.exit(run(args))


if __name__ == "__main__":
    main()
