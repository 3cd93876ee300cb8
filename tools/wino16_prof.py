#!/usr/bin/env python3
"""A few launches of the wide decoder's Winograd kernels, for rocprofv3 counter passes.
    rocprofv3 --pmc SQ_WAVE_CYCLES ... -- python3 tools/wino16_prof.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops  # noqa: E402

dev = torch.device("cuda")
B = 16
g = torch.Generator(device="cpu").manual_seed(2)
x = torch.relu(torch.randn(B, 16, 35, 35, 35, generator=g) * 0.7).to(dev)
gy = (torch.randn(B, 16, 32, 32, 32, generator=g) * (torch.rand(B, 16, 32, 32, 32, generator=g) < 0.6)).to(dev)
w = (torch.randn(16, 16, 4, 4, 4, generator=g) * 0.06).to(dev)
wf, wb = ops.pack_conv_weight(w)
b = torch.zeros(16, device=dev)
wbat = ops.WgradBatch(dev)
out = torch.empty(16 * 16 * 64, device=dev)
for _ in range(5):
    base = wbat.reserve(256 * 16384 * 4)
    n = ops.wgrad16_k4_wino_partial(gy, x, base)
    wbat.add_job(base, out, n, 16384)
    wbat.finish()
    ops.conv3d_k4_wino16_bwd(gy, ops.pack_wino16_k4(wb), x)
    ops.conv3d_k4_wino16_fwd(x, ops.pack_wino16_k4(wf), b)
torch.cuda.synchronize()
