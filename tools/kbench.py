#!/usr/bin/env python3
"""Per-kernel timing of the decoder's layer ops on the GPU (HIP events), for tuning.

    python tools/kbench.py --batch 16 --variants 0,2,3,4
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--variants", default="0")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--chanstr", default="8,16,8,8")
    ap.add_argument("--ch", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--check", action="store_true", help="also compare every variant bit for bit with variant 1")
    a = ap.parse_args()
    B = a.batch
    c0, c1, c2, c3 = (int(v) for v in a.chanstr.split(","))
    dev = torch.device("cuda")
    R = lambda *s: torch.randn(*s, device=dev)
    variants = [int(v) for v in a.variants.split(",")]
    cases = []

    def conv(name, cin, cout, k, pad, n):
        no = n + 2 * pad - k + 1
        x, w, b = R(B, cin, n, n, n), R(cout, cin, k, k, k), R(cout)
        wf, wb = ops.pack_conv_weight(w)
        gy = R(B, cout, no, no, no)
        macs = B * cout * no ** 3 * cin * k ** 3
        cases.append((name + ".fwd", macs, lambda: ops.conv3d_gather(x, wf, b, cout, k, 1, pad, (no, no, no), 1)))
        cases.append((name + ".bwd_data", macs, lambda: ops.conv3d_gather(gy, wb, None, cin, k, 1, k - 1 - pad, (n, n, n), mask=x)))
        cases.append((name + ".bwd_weight", macs, lambda: ops.wgrad(gy, x, k, 1, pad, out_mode=0)))
        if cout == 1:
            cases.append((name + ".bwd_weight_flip", macs, lambda: ops.wgrad(x, gy, k, 1, k - 1 - pad, out_mode=1)))

    def convT(name, cin, cout, pad, n):
        x, w, b = R(B, cin, n, n, n), R(cin, cout, 5, 5, 5), R(cout)
        wf, wb = ops.pack_convT_weight(w)
        no = 2 * n + (3 if pad == 0 else 0)
        gy = R(B, cout, no, no, no)
        macs = B * cin * n ** 3 * cout * 125
        cases.append((name + ".fwd", macs, lambda: ops.convT3d_k5s2_fwd(x, wf, b, cout, pad, 1)))
        cases.append((name + ".bwd_data", macs, lambda: ops.conv3d_gather(gy, wb, None, cin, 5, 2, pad, (n, n, n), mask=x)))
        cases.append((name + ".bwd_weight", macs, lambda: ops.wgrad(x, gy, 5, 2, pad, out_mode=0)))

    conv("conv2", c3, c3, 4, 0, 35)
    convT("up2", c2, c3, 0, 16)
    conv("conv1", c2, c2, 4, 0, 19)
    convT("up1", c1, c2, 0, 8)
    conv("cls2", c3, 1, 3, 1, 32)
    conv("cls1", c2, 1, 3, 1, 16)
    conv("cls0", c1, 1, 3, 1, 8)
    convT("conv0", c0, c1, 2, 4)
    convT("up0", a.ch, c0, 2, 2)
    x35 = R(B, c3, 35, 35, 35)
    cases.append(("channel_sum[35^3]", 0, lambda: ops.channel_sum(x35)))
    print(f"batch {B}, chanstr {a.chanstr}")
    print(f"{'op':22s}" + "".join(f"{'v' + str(v):>22s}" for v in variants))
    total = {v: 0.0 for v in variants}
    for name, macs, fn in cases:
        if a.only and a.only not in name:
            continue
        row = f"{name:22s}"
        ref = None
        if a.check:
            ops.set_variant(1)
            ref = fn()
        for v in variants:
            ops.set_variant(v)
            if ref is not None:
                got = fn()
                torch.cuda.synchronize()
                if not torch.equal(got, ref):
                    row += f" [v{v} MISMATCH max|d|={float((got - ref).abs().max()):.3e}]"
            us = timeit(fn, a.reps)
            total[v] += us
            row += f"{us:10.1f}us {2 * macs / us / 1e6:6.1f}TF  " if macs else f"{us:10.1f}us            "
        print(row, flush=True)
    ops.set_variant(0)
    print("sum".ljust(22) + "".join(f"{total[v]:10.1f}us            " for v in variants))


if __name__ == "__main__":
    main()
