#!/usr/bin/env python3
"""Times the 16-row matrix-core gather convolutions of the wide decoder (conv16_mfma.hip) over their tile variants.

    python tools/g16_sweep.py [--batch 16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops  # noqa: E402
from tools.trunk_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--variants", default="0,2,3,4,5,6,7,8,9,10,11,12,13")
    a = ap.parse_args()
    B, dev = a.batch, torch.device("cuda")
    R = lambda *s: torch.randn(*s, device=dev)
    cases = []

    def conv(name, n):
        w = R(16, 16, 4, 4, 4) * 0.05
        wf, wb = ops.pack_conv_weight(w)
        wpf, wpb = ops.pack_g16_mfma(wf, 16, 16, 4), ops.pack_g16_mfma(wb, 16, 16, 4)
        no = n - 3
        x, gy, b = R(B, 16, n, n, n), R(B, 16, no, no, no), R(16)
        y, dx = torch.empty(B, 16, no, no, no, device=dev), torch.empty(B, 16, n, n, n, device=dev)
        macs = B * 16 * no ** 3 * 16 * 64
        cases.append((name + ".fwd", macs, lambda v: ops.conv3d_g16_mfma(x, wpf, b, 16, 4, 1, 0, (no, no, no), ops.ACT_RELU, out=y, variant=v)))
        cases.append((name + ".bwd_data", macs, lambda v: ops.conv3d_g16_mfma(gy, wpb, None, 16, 4, 1, 3, (n, n, n), mask=x, out=dx, variant=v)))

    def convT_bwd(name, cin, cout, n, pad):
        w = R(cin, cout, 5, 5, 5) * 0.05
        wf, wb = ops.pack_convT_weight(w)
        wpb = ops.pack_g16_mfma(wb, cout, cin, 5)
        no = 2 * n + (3 if pad == 0 else 0)
        x, gy = R(B, cin, n, n, n), R(B, cout, no, no, no)
        dx = torch.empty(B, cin, n, n, n, device=dev)
        macs = B * cin * n ** 3 * cout * 125
        cases.append((name + ".bwd_data", macs, lambda v: ops.conv3d_g16_mfma(gy, wpb, None, cin, 5, 2, pad, (n, n, n), mask=x, out=dx, variant=v)))

    conv("conv2", 35)
    conv("conv1", 19)
    convT_bwd("up2", 16, 16, 16, 0)
    convT_bwd("up1", 32, 16, 8, 0)
    convT_bwd("conv0", 16, 32, 4, 2)
    for name, macs, fn in cases:
        line = f"{name:16s}"
        for v in [int(t) for t in a.variants.split(",")]:
            try:
                us = timeit(lambda: fn(v), reps=10)
                line += f"  v{v}: {us:7.1f} ({2 * macs / us / 1e6:5.1f} TF)"
            except RuntimeError:
                pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
