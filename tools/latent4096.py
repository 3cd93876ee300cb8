#!/usr/bin/env python3
"""BASELINE.json configs[2]: 4096 synthetic 32^3 blocks resident on one GPU, the full-batch latent step
(NVFPCC.py:225-251: forward mode 'train' q = 1, losses, backward-data through the whole decoder, Adam on the latent
table; no weight gradients) and the eval forward, for the rocprofv3 HBM-traffic passes (tools/profile_4096.sh).

    python3 tools/latent4096.py [--blocks 4096] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    from nvfpcc_amd import network
    from nvfpcc_amd.engine import TrainEngine
    from nvfpcc_amd.model import Net
    from nvfpcc_amd.seeds import synthetic_seed
    from nvfpcc_amd.synth import make_blocks
    dev = torch.device("cuda", 0)
    network.reset_seed(synthetic_seed())
    net = Net(None, "Gaussian", 3, "8,16,8,8", verbose=False).to(dev)
    gts, dists = make_blocks(128)
    reps = (a.blocks + 127) // 128
    gt = torch.from_numpy(np.tile(gts, (reps, 1, 1, 1, 1))[:a.blocks]).float().to(dev)
    dist = torch.from_numpy(np.tile(dists, (reps, 1, 1, 1, 1))[:a.blocks]).float().to(dev)
    eng = TrainEngine(net, gt, dist, n_points_total=float(gt.sum().item()), lmbda=200.0, w1=10.0, w2=57.0, lr=1e-3,
                      wemb=5.0, seed=0)
    eng.latent_step(1)                                   # warm-up: workspaces
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        eng.latent_step(1)
    torch.cuda.synchronize()
    dt_lat = (time.perf_counter() - t0) / a.reps
    eng.eval_forward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        eng.eval_forward()
    torch.cuda.synchronize()
    dt_eval = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"blocks": a.blocks, "latent_step_ms": round(dt_lat * 1e3, 2),
                      "latent_step_blocks_per_s": round(a.blocks / dt_lat, 1),
                      "latent_step_tflops": round(a.blocks / dt_lat * 0.8048e9 / 1e12, 2),
                      "eval_forward_ms": round(dt_eval * 1e3, 2),
                      "eval_forward_blocks_per_s": round(a.blocks / dt_eval, 1),
                      "eval_forward_tflops": round(a.blocks / dt_eval * 0.4024e9 / 1e12, 2),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
