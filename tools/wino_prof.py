#!/usr/bin/env python3
"""A few launches of one Winograd kernel, for rocprofv3 counter passes.
    rocprofv3 --pmc SQ_WAVE_CYCLES ... -- python3 tools/wino_prof.py --case wgrad --zsplit 2
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="wgrad")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--zsplit", type=int, default=2)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda")
B = a.batch
g = torch.Generator(device="cpu").manual_seed(2)
x = torch.relu(torch.randn(B, 8, 35, 35, 35, generator=g) * 0.7).to(dev)
gy = (torch.randn(B, 8, 32, 32, 32, generator=g) * (torch.rand(B, 8, 32, 32, 32, generator=g) < 0.6)).to(dev)
if a.case == "wgrad":
    for _ in range(a.reps):
        ops.wgrad_k4_wino(gy, x, zsplit=a.zsplit)
else:
    w = (torch.randn(8, 8, 4, 4, 4, generator=g) * 0.08).to(dev)
    _, wb = ops.pack_conv_weight(w)
    wp = ops.pack_wino_k4(wb)
    for _ in range(a.reps):
        ops.conv3d_k4_wino_bwd(gy, wp, x)
torch.cuda.synchronize()
