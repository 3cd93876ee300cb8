#!/usr/bin/env python3
"""conv1 backward-data (8 -> 8, k4, 16^3 -> 19^3): every matrix-core mapping / variant, timed standalone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops
from tools.trunk_bench import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
w = torch.randn(8, 8, 4, 4, 4, device=dev) * 0.05
wf, wb = ops.pack_conv_weight(w)
x, gy = torch.randn(B, 8, 19, 19, 19, device=dev), torch.randn(B, 8, 16, 16, 16, device=dev)
dx = torch.empty_like(x)
ref = None
for pair in (2, 0):
    wp = ops.pack_mfma_k4(wb, 8, pair)
    for var in (0, 2, 3):
        ops.set_mfma_variant(var)
        try:
            us = timeit(lambda: ops.conv3d_k4_mfma(gy, wp, None, 3, pair, mask=x, out=dx))
        except RuntimeError as e:
            print(f"pair {pair} variant {var}: n/a")
            continue
        if ref is None:
            ref = dx.clone()
        print(f"pair {pair} variant {var}: {us:7.1f} us   same bits as first: {torch.equal(dx, ref)}", flush=True)
