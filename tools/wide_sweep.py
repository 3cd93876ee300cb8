#!/usr/bin/env python3
"""Times the VALU tile kernels of the wide decoder (chanstr 16,32,16,16) standalone over their tuning candidates
(variant ids 40-43 in conv_direct.hip / wgrad.hip; 0 = the shipped choice).

    python tools/wide_sweep.py [--batch 16]
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
    ap.add_argument("--variants", default="0,40,41,42,43")
    a = ap.parse_args()
    B, dev = a.batch, torch.device("cuda")
    R = lambda *s: torch.randn(*s, device=dev)
    cases = []

    def convT(name, cin, cout, n):
        w = R(cin, cout, 5, 5, 5) * 0.05
        wf, wb = ops.pack_convT_weight(w)
        no = 2 * n + 3
        x, gy = R(B, cin, n, n, n), R(B, cout, no, no, no)
        y = torch.empty(B, cout, no, no, no, device=dev)
        macs = B * cin * n ** 3 * cout * 125
        cases.append((name + ".fwd", macs, lambda: ops.convT3d_k5s2_fwd(x, wf, None, cout, 0, ops.ACT_RELU, out=y)))
        if cout == 16:
            wp16 = ops.pack_convT16_mfma(wf, cin, 16)
            for v in (0, 2, 3):
                cases.append((name + f".fwd.mfma16.v{v}", macs,
                              lambda v=v: ops.convT3d_k5s2_mfma16(x, wp16, None, ops.ACT_RELU, out=y, variant=v)))
        dx = torch.empty(B, cin, n, n, n, device=dev)
        cases.append((name + ".bwd_data", macs,
                      lambda: ops.conv3d_gather(gy, wb, None, cin, 5, 2, 0, (n, n, n), mask=x, out=dx)))
        dw = torch.empty(cin, cout, 5, 5, 5, device=dev)
        cases.append((name + ".wgrad", macs, lambda: ops.wgrad(x, gy, 5, 2, 0, out=dw)))

    def convT_p2(name, cin, cout, n):          # padding 2, output_padding 1: n -> 2n (conv0)
        w = R(cin, cout, 5, 5, 5) * 0.05
        wf, wb = ops.pack_convT_weight(w)
        no = 2 * n
        x, gy = R(B, cin, n, n, n), R(B, cout, no, no, no)
        y = torch.empty(B, cout, no, no, no, device=dev)
        macs = B * cin * n ** 3 * cout * 125
        cases.append((name + ".fwd", macs, lambda: ops.convT3d_k5s2_fwd(x, wf, None, cout, 2, ops.ACT_RELU, out=y)))
        dx = torch.empty(B, cin, n, n, n, device=dev)
        cases.append((name + ".bwd_data", macs,
                      lambda: ops.conv3d_gather(gy, wb, None, cin, 5, 2, 2, (n, n, n), out=dx)))

    def conv(name, c, n):
        w = R(c, c, 4, 4, 4) * 0.05
        wf, wb = ops.pack_conv_weight(w)
        no = n - 3
        x, gy = R(B, c, n, n, n), R(B, c, no, no, no)
        y, dx = torch.empty(B, c, no, no, no, device=dev), torch.empty(B, c, n, n, n, device=dev)
        macs = B * c * no ** 3 * c * 64
        cases.append((name + ".fwd", macs, lambda: ops.conv3d_gather(x, wf, None, c, 4, 1, 0, (no, no, no), ops.ACT_RELU, out=y)))
        cases.append((name + ".bwd_data", macs, lambda: ops.conv3d_gather(gy, wb, None, c, 4, 1, 3, (n, n, n), mask=x, out=dx)))
        dw = torch.empty(c, c, 4, 4, 4, device=dev)
        cases.append((name + ".wgrad", macs, lambda: ops.wgrad(gy, x, 4, 1, 0, out=dw)))

    convT_p2("conv0", 16, 32, 4)
    convT("up1", 32, 16, 8)
    conv("conv1", 16, 19)
    convT("up2", 16, 16, 16)
    conv("conv2", 16, 35)
    for name, macs, fn in cases:
        line = f"{name:16s}"
        for v in [int(t) for t in a.variants.split(",")]:
            ops.set_variant(v)
            try:
                us = timeit(fn, reps=10)
                line += f"  v{v}: {us:8.1f} us ({2 * macs / us / 1e6:5.1f} TF)"
            except RuntimeError:
                line += f"  v{v}: n/a"
        ops.set_variant(0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
