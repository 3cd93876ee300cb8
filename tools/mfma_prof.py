#!/usr/bin/env python3
"""Run one matrix-core convolution a few times (for rocprofv3 counter passes).

    rocprofv3 --pmc SQ_WAVE_CYCLES ... -- python3 tools/mfma_prof.py --batch 16 --case conv2.fwd --variant 0
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops  # noqa: E402

CASES = {"conv2.fwd": (35, 0, 0), "conv1.fwd": (19, 0, 0), "conv2.bwd_data": (32, 3, 2), "conv1.bwd_data": (16, 3, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--case", default="conv2.fwd")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    n, pad, pair = CASES[a.case]
    dev = torch.device("cuda")
    x = torch.randn(a.batch, 8, n, n, n, device=dev)
    w = torch.randn(8, 8, 4, 4, 4, device=dev) * 0.05
    wf, wb = ops.pack_conv_weight(w)
    wp = ops.pack_mfma_k4(wf if pad == 0 else wb, 8, pair)
    ops.set_mfma_variant(a.variant)
    for _ in range(a.reps):
        ops.conv3d_k4_mfma(x, wp, None, pad, pair, ops.ACT_RELU)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
