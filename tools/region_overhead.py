"""Where do the ~75 us of a 20-step timed region that are not step time go?  Wall time of regions made of different graph
replays (sizes in replay order), minus sizes x the steady ms/step."""
import argparse, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from nvfpcc_amd.engine import GraphedTrainStep

args = bench.parse_args() if hasattr(bench, "parse_args") else None
args.blocks, args.distinct = 917, 128
dev = torch.device("cuda")
eng = bench.build_engine(args, dev, 1)
g = GraphedTrainStep(eng, 16, 1, unroll=(20, 16, 8, 4, 2))
g.prime(4)
rng = np.random.default_rng(0)


def region(sizes, reps=7):
    n = sum(sizes)
    out = []
    for _ in range(reps):
        ids = np.stack([rng.permutation(917)[:16] for _ in range(n)]).astype(np.int64)
        h = g.stage_schedule((ids, eng.counts[ids].sum(axis=1)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.load_schedule(h)
        t1 = time.perf_counter()
        for u in sizes:
            if u == 1:
                g.replay()
            else:
                gr, out_, last = g.graphs_u[u]
                del g.pending[:u]
                eng.noise_step += u; eng.opt_step += u
                gr.replay()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        out.append(((t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6))
    out.sort()
    return out[len(out) // 2]


step = (region([16] * 8)[0] - region([16] * 4)[0]) / 64
print(f"steady step {step:.2f} us")
region([1, 4], reps=1)
tot, tl, tr = region([20], reps=1)
print(f"[20] right after [1, 4] (the bench's first region): + {tot - 20 * step:6.1f} us fixed")
for sizes in ([20], [4] * 5, [2] * 10, [20], [4] * 5, [2] * 10, [1] * 20):
    tot, tl, tr = region(sizes)
    print(f"{str(sizes):18s} total {tot:8.1f} us = {sum(sizes)} steps + {tot - sum(sizes) * step:6.1f} us fixed   (host: load_schedule {tl:5.1f}, replays {tr:6.1f})")
