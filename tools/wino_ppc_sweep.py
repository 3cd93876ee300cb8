#!/usr/bin/env python3
"""Sweep the one-set Winograd kernels' z-pairs-per-chunk (ppc) at a large batch: the batch-16 defaults trade redundant plane
transforms for workgroup count, which a batch-917 launch does not need.

    python tools/wino_ppc_sweep.py --batch 256
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops  # noqa: E402
from tools.wino_bench import timeit  # noqa: E402

ONE = 1 << 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    B, dev = a.batch, torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(3)
    w = (torch.randn(8, 8, 4, 4, 4, generator=g) * 0.08).to(dev)
    b = (torch.randn(8, generator=g) * 0.1).to(dev)
    wf, wb = ops.pack_conv_weight(w)
    wwf, wwb = ops.pack_wino_k4(wf), ops.pack_wino_k4(wb)
    for n, ppcs in ((35, (2, 4, 8, 16)), (19, (1, 2, 4, 8))):
        x = torch.relu(torch.randn(B, 8, n, n, n, device=dev))
        out = torch.empty(B, 8, n - 3, n - 3, n - 3, device=dev)
        ref = None
        for ppc in ppcs:
            y = ops.conv3d_k4_wino_fwd(x, wwf, b, ppc=ONE | ppc)
            ref = y.clone() if ref is None else ref
            us = timeit(lambda: ops.conv3d_k4_wino_fwd(x, wwf, b, out=out, ppc=ONE | ppc), a.reps)
            print(f"fwd {n} ppc {ppc:2d}: {us:8.1f} us   same bits as first: {bool(torch.equal(y, ref))}", flush=True)
        gy = torch.randn(B, 8, n - 3, n - 3, n - 3, device=dev)
        dx = torch.empty(B, 8, n, n, n, device=dev)
        ref = None
        for ppc in ((3, 6, 9, 18) if n == 35 else (1, 2, 5, 10)):
            y = ops.conv3d_k4_wino_bwd(gy, wwb, x, ppc=ONE | ppc)
            ref = y.clone() if ref is None else ref
            us = timeit(lambda: ops.conv3d_k4_wino_bwd(gy, wwb, x, out=dx, ppc=ONE | ppc), a.reps)
            print(f"bwd-data {n} ppc {ppc:2d}: {us:8.1f} us   same bits as first: {bool(torch.equal(y, ref))}", flush=True)


if __name__ == "__main__":
    main()
