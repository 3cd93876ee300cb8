#!/usr/bin/env python3
"""Matrix-core 4^3 convolutions vs the VALU kernels: max error and time per variant (tuning aid).

    python tools/mfma_check.py --batch 16
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--variants", default="0,2,3,4,5")
    a = ap.parse_args()
    B = a.batch
    dev = torch.device("cuda")
    torch.manual_seed(0)
    for name, n, pad, pair in (("conv2.fwd", 35, 0, 0), ("conv1.fwd", 19, 0, 0), ("conv2.bwd_data", 32, 3, 2),
                               ("conv1.bwd_data", 16, 3, 2), ("conv2.bwd_flat", 32, 3, 0), ("conv1.bwd_flat", 16, 3, 0)):
        no = n + 2 * pad - 3
        x = torch.randn(B, 8, n, n, n, device=dev)
        w = torch.randn(8, 8, 4, 4, 4, device=dev) * 0.05
        bias = torch.randn(8, device=dev)
        wf, wb = ops.pack_conv_weight(w)
        gw = wf if pad == 0 else wb
        mask = torch.randn(B, 8, no, no, no, device=dev) if pad else None
        act = ops.ACT_RELU if pad == 0 else ops.ACT_NONE
        bb = bias if pad == 0 else None
        ref = ops.conv3d_gather(x, gw, bb, 8, 4, 1, pad, (no, no, no), act, mask=mask)
        t_ref = timeit(lambda: ops.conv3d_gather(x, gw, bb, 8, 4, 1, pad, (no, no, no), act, mask=mask))
        macs = B * 8 * (min(n, no) ** 3) * 8 * 64
        wp = ops.pack_mfma_k4(gw, 8, pair)
        row = f"{name:16s} valu {t_ref:8.1f}us {2 * macs / t_ref / 1e6:6.1f}TF |"
        for v in [int(v) for v in a.variants.split(",")]:
            ops.set_mfma_variant(v)
            try:
                got = ops.conv3d_k4_mfma(x, wp, bb, pad, pair, act, mask=mask)
            except Exception:
                continue
            torch.cuda.synchronize()
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            t = timeit(lambda: ops.conv3d_k4_mfma(x, wp, bb, pad, pair, act, mask=mask))
            row += f" v{v}: {t:7.1f}us {2 * macs / t / 1e6:6.1f}TF err {err:.1e} |"
        print(row, flush=True)
    for name, cin, n in (("up2.fwd", 8, 16), ("up1.fwd", 16, 8)):
        x = torch.randn(B, cin, n, n, n, device=dev)
        w = torch.randn(cin, 8, 5, 5, 5, device=dev) * 0.05
        bias = torch.randn(8, device=dev)
        wf, _ = ops.pack_convT_weight(w)
        ops.set_mfma_variant(0)
        ref = ops.convT3d_k5s2_fwd(x, wf, bias, 8, 0, ops.ACT_RELU)
        t_ref = timeit(lambda: ops.convT3d_k5s2_fwd(x, wf, bias, 8, 0, ops.ACT_RELU))
        macs = B * cin * n ** 3 * 8 * 125
        wp = ops.pack_convT_mfma(wf, cin)
        row = f"{name:16s} valu {t_ref:8.1f}us {2 * macs / t_ref / 1e6:6.1f}TF |"
        for v in [int(v) for v in a.variants.split(",")]:
            ops.set_mfma_variant(v)
            try:
                got = ops.convT3d_k5s2_mfma(x, wp, bias, ops.ACT_RELU)
            except Exception:
                continue
            torch.cuda.synchronize()
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            t = timeit(lambda: ops.convT3d_k5s2_mfma(x, wp, bias, ops.ACT_RELU))
            row += f" v{v}: {t:7.1f}us {2 * macs / t / 1e6:6.1f}TF err {err:.1e} |"
        print(row, flush=True)
    for name, cin, n in (("up2.bwd_data", 8, 16), ("up1.bwd_data", 16, 8)):
        no = 2 * n + 3
        gy = torch.randn(B, 8, no, no, no, device=dev)
        w = torch.randn(cin, 8, 5, 5, 5, device=dev) * 0.05
        _, wb = ops.pack_convT_weight(w)
        mask = torch.randn(B, cin, n, n, n, device=dev)
        add = torch.randn(B, cin, n, n, n, device=dev)
        ops.set_mfma_variant(0)
        ref = ops.conv3d_gather(gy, wb, None, cin, 5, 2, 0, (n, n, n), addend=add, mask=mask)
        t_ref = timeit(lambda: ops.conv3d_gather(gy, wb, None, cin, 5, 2, 0, (n, n, n), addend=add, mask=mask))
        macs = B * cin * n ** 3 * 8 * 125
        wp = ops.pack_s2k5_mfma(wb, 8, cin)
        row = f"{name:16s} valu {t_ref:8.1f}us {2 * macs / t_ref / 1e6:6.1f}TF |"
        for v in [int(v) for v in a.variants.split(",")]:
            ops.set_mfma_variant(v)
            try:
                got = ops.conv3d_s2k5_mfma(gy, wp, cin, addend=add, mask=mask)
            except Exception:
                continue
            torch.cuda.synchronize()
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            t = timeit(lambda: ops.conv3d_s2k5_mfma(gy, wp, cin, addend=add, mask=mask))
            row += f" v{v}: {t:7.1f}us {2 * macs / t / 1e6:6.1f}TF err {err:.1e} |"
        print(row, flush=True)
    ops.set_mfma_variant(0)


if __name__ == "__main__":
    main()
