#!/usr/bin/env python3
"""Times every matrix-core kernel of the narrow trunk (chanstr 8,16,8,8) standalone, outputs preallocated.

    python tools/trunk_bench.py [--batch 16] [--variant 0] [--only up2]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    B, dev = a.batch, torch.device("cuda")
    R = lambda *s: torch.randn(*s, device=dev)
    ops.set_mfma_variant(a.variant)
    cases = []

    def conv(name, n):                         # 8 -> 8, k4, valid: n -> n - 3
        w = R(8, 8, 4, 4, 4) * 0.05
        wf, wb = ops.pack_conv_weight(w)
        no = n - 3
        x, gy = R(B, 8, n, n, n), R(B, 8, no, no, no)
        wpf = ops.pack_mfma_k4(wf, 8, 0)
        y, dx = torch.empty(B, 8, no, no, no, device=dev), torch.empty(B, 8, n, n, n, device=dev)
        macs = B * 8 * no ** 3 * 8 * 64
        cases.append((name + ".fwd", macs, lambda: ops.conv3d_k4_mfma(x, wpf, None, 0, 0, ops.ACT_RELU, out=y)))
        from nvfpcc_amd.engine import MFMA_BWD
        pair = MFMA_BWD[name][0]
        wpb = ops.pack_mfma_k4(wb, 8, pair)
        cases.append((name + ".bwd_data", macs, lambda: ops.conv3d_k4_mfma(gy, wpb, None, 3, pair, mask=x, out=dx)))
        wpb2 = ops.pack_mfma_k4(wb, 8, 2 - pair)            # the other pair axis, and the VALU tile kernel
        cases.append((name + f".bwd_data.pair{2 - pair}", macs,
                      lambda: ops.conv3d_k4_mfma(gy, wpb2, None, 3, 2 - pair, mask=x, out=dx)))
        cases.append((name + ".bwd_data.valu", macs,
                      lambda: ops.conv3d_gather(gy, wb, None, 8, 4, 1, 3, (n, n, n), mask=x)))

    def convT(name, cin, n):                   # cin -> 8, k5 s2, padding 0: n -> 2n + 3
        w = R(cin, 8, 5, 5, 5) * 0.05
        wf, wb = ops.pack_convT_weight(w)
        no = 2 * n + 3
        x, gy = R(B, cin, n, n, n), R(B, 8, no, no, no)
        wpt, wps = ops.pack_convT_mfma(wf, cin), ops.pack_s2k5_mfma(wb, 8, cin)
        y, dx = torch.empty(B, 8, no, no, no, device=dev), torch.empty(B, cin, n, n, n, device=dev)
        macs = B * cin * n ** 3 * 8 * 125
        cases.append((name + ".fwd", macs, lambda: ops.convT3d_k5s2_mfma(x, wpt, None, ops.ACT_RELU, out=y)))
        cases.append((name + ".bwd_data", macs, lambda: ops.conv3d_s2k5_mfma(gy, wps, cin, mask=x, out=dx)))

    conv("conv2", 35)
    convT("up2", 8, 16)
    conv("conv1", 19)
    convT("up1", 16, 8)
    for name, macs, fn in cases:
        if a.only and a.only not in name:
            continue
        try:
            us = timeit(fn)
        except RuntimeError:
            print(f"{name:18s}      n/a (no such variant)", flush=True)
            continue
        print(f"{name:18s} {us:8.1f} us  {2 * macs / us / 1e6:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
