#!/usr/bin/env python3
"""Experiment: how much does running K sub-batches of the train step on K streams concurrently buy at a fixed
total batch?  K independent engines (own parameters -- this is a scheduling probe, not a correct trainer)."""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nvfpcc_amd.engine import GraphedTrainStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--total", type=int, default=16)
    ap.add_argument("--ks", default="1,2,4")
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for K in [int(k) for k in a.ks.split(",")]:
        b = a.total // K
        args = SimpleNamespace(ch=3, chanstr="8,16,8,8", blocks=256, distinct=64, force_collective=False)
        engs = [bench.build_engine(args, dev, 1) for _ in range(K)]
        streams = [torch.cuda.Stream() for _ in range(K)]
        graphed = []
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                graphed.append(GraphedTrainStep(e, b, 1))
        torch.cuda.synchronize()
        ids = np.arange(b)

        def step():
            for g, s in zip(graphed, streams):
                with torch.cuda.stream(s):
                    g(ids, n_pts=1000.0)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"K={K} sub-batch={b}: {dt * 1e3:.3f} ms/step  {a.total / dt:.0f} blocks/s", flush=True)
        del graphed, engs


if __name__ == "__main__":
    main()
