#!/usr/bin/env python3
"""A/B of conv2's backward-data at the bench batch: the direct matrix-core form (conv_k4_mfma, flattened x-pair) against
the Winograd (y, x) form (conv_wino.hip).  HIP events around `reps` back-to-back launches, L2-cold-ish inputs rotated.

    python tools/wino_bench.py --batch 16 --ppc 6,2,18
"""
import argparse
import ctypes
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
    return s.elapsed_time(e) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--ppc", default="6")
    a = ap.parse_args()
    B = a.batch
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(1)
    w = (torch.randn(8, 8, 4, 4, 4, generator=g) * 0.08).to(dev)
    gy = (torch.randn(B, 8, 32, 32, 32, generator=g) * (torch.rand(B, 8, 32, 32, 32, generator=g) < 0.6)).to(dev)
    mask = torch.randn(B, 8, 35, 35, 35, generator=g).to(dev)
    _, wb = ops.pack_conv_weight(w)
    wp_d, wp_w = ops.pack_mfma_k4(wb, 8, 0), ops.pack_wino_k4(wb)
    slabs = torch.zeros(4096 * 8, device=dev)
    out = torch.empty(B, 8, 35, 35, 35, device=dev)
    macs = B * 8 * 8 * 64 * 32 ** 3
    d = ops.conv3d_k4_mfma(gy, wp_d, None, 3, 0, ops.ACT_NONE, mask=mask)
    us = timeit(lambda: ops.conv3d_k4_mfma(gy, wp_d, None, 3, 0, ops.ACT_NONE, mask=mask, out=out,
                                           bias_part=slabs.data_ptr() if B <= 32 else None), a.reps)
    print(f"direct  (conv_k4_mfma flat x-pair): {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF algorithmic")
    for ppc in [int(v) for v in a.ppc.split(",")]:
        wv = ops.conv3d_k4_wino_bwd(gy, wp_w, mask, ppc=ppc)
        err = float((wv - d).abs().max() / d.abs().max()) if ppc < 256 else float("nan")
        # (the slab buffer holds 4096 units' sums: enough for batch <= 32 only -- beyond that an out-of-bounds write)
        bp = slabs.data_ptr() if B <= 32 else None
        us = timeit(lambda: ops.conv3d_k4_wino_bwd(gy, wp_w, mask, out=out, bias_part=bp, ppc=ppc), a.reps)
        print(f"winograd (conv_k4_wino_bwd, ppc {ppc & 255:2d} dbg {ppc >> 8:2d}): {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF algorithmic   max|d| / max = {err:.2e}")


def fwd_main(B, reps):
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(3)
    for n in (35, 19):
        w = (torch.randn(8, 8, 4, 4, 4, generator=g) * 0.08).to(dev)
        b = (torch.randn(8, generator=g) * 0.1).to(dev)
        x = torch.relu(torch.randn(B, 8, n, n, n, generator=g)).to(dev)
        wf, wb = ops.pack_conv_weight(w)
        wp_d, wp_w = ops.pack_mfma_k4(wf, 8, 0), ops.pack_wino_k4(wf)
        macs = B * 8 * 8 * 64 * (n - 3) ** 3
        out = torch.empty(B, 8, n - 3, n - 3, n - 3, device=dev)
        d = ops.conv3d_k4_mfma(x, wp_d, b, 0, 0, ops.ACT_RELU)
        us = timeit(lambda: ops.conv3d_k4_mfma(x, wp_d, b, 0, 0, ops.ACT_RELU, out=out), reps)
        print(f"fwd {n} direct:            {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF")
        for ppc in ((4, 2, 6, 8, 16, 65538, 65540, 65544, 65537) if n == 35 else (2, 4, 8, 65537, 65538)):   # bit 16: conv_wino1.hip
            y = ops.conv3d_k4_wino_fwd(x, wp_w, b, ppc=ppc)
            err = float((y - d).abs().max() / d.abs().max())
            us = timeit(lambda: ops.conv3d_k4_wino_fwd(x, wp_w, b, out=out, ppc=ppc), reps)
            print(f"fwd {n} winograd ppc {ppc}:    {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF   max|d| / max = {err:.2e}")
    # conv1 backward-data
    w = (torch.randn(8, 8, 4, 4, 4, generator=g) * 0.08).to(dev)
    gy = torch.randn(B, 8, 16, 16, 16, generator=g).to(dev)
    mask = torch.randn(B, 8, 19, 19, 19, generator=g).to(dev)
    _, wb = ops.pack_conv_weight(w)
    wp_d, wp_w = ops.pack_mfma_k4(wb, 8, 2), ops.pack_wino_k4(wb)
    macs = B * 8 * 8 * 64 * 16 ** 3
    d = ops.conv3d_k4_mfma(gy, wp_d, None, 3, 2, ops.ACT_NONE, mask=mask)
    us = timeit(lambda: ops.conv3d_k4_mfma(gy, wp_d, None, 3, 2, ops.ACT_NONE, mask=mask), reps)
    print(f"conv1 bwd-data direct (z pair): {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF")
    for ppc in (2, 4, 6, 10, 65537, 65538, 65541):
        y = ops.conv3d_k4_wino_bwd(gy, wp_w, mask, ppc=ppc)
        err = float((y - d).abs().max() / d.abs().max())
        us = timeit(lambda: ops.conv3d_k4_wino_bwd(gy, wp_w, mask, ppc=ppc), reps)
        print(f"conv1 bwd-data winograd ppc {ppc:2d}: {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF   max|d| / max = {err:.2e}")


def wide_main(B, reps):
    """The wide decoder's 4^3 layers (16 -> 16 channels): direct 16-row kernel against conv16_wino.hip."""
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(5)
    for n in (35, 19):
        no = n - 3
        w = (torch.randn(16, 16, 4, 4, 4, generator=g) * 0.06).to(dev)
        b = (torch.randn(16, generator=g) * 0.1).to(dev)
        x = torch.relu(torch.randn(B, 16, n, n, n, generator=g)).to(dev)
        gy = (torch.randn(B, 16, no, no, no, generator=g) * (torch.rand(B, 16, no, no, no, generator=g) < 0.6)).to(dev)
        wf, wb = ops.pack_conv_weight(w)
        macs = B * 16 * 16 * 64 * no ** 3
        out = torch.empty(B, 16, no, no, no, device=dev)
        dxo = torch.empty(B, 16, n, n, n, device=dev)
        wpf, wpb = ops.pack_g16_mfma(wf, 16, 16, 4), ops.pack_g16_mfma(wb, 16, 16, 4)
        d = ops.conv3d_g16_mfma(x, wpf, b, 16, 4, 1, 0, (no, no, no), ops.ACT_RELU)
        us = timeit(lambda: ops.conv3d_g16_mfma(x, wpf, b, 16, 4, 1, 0, (no, no, no), ops.ACT_RELU, out=out), reps)
        print(f"wide fwd {n} direct:            {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF")
        ww = ops.pack_wino16_k4(wf)
        for ppc in ((4, 2, 8, 16, 65540, 65538, 65544) if n == 35 else (2, 1, 4, 8)):     # bit 16: conv16_wino1.hip
            y = ops.conv3d_k4_wino16_fwd(x, ww, b, ppc=ppc)
            err = float((y - d).abs().max() / d.abs().max())
            us = timeit(lambda: ops.conv3d_k4_wino16_fwd(x, ww, b, out=out, ppc=ppc), reps)
            print(f"wide fwd {n} winograd ppc {ppc:2d}:   {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF   max|d| / max = {err:.2e}")
        d = ops.conv3d_g16_mfma(gy, wpb, None, 16, 4, 1, 3, (n, n, n), mask=x)
        us = timeit(lambda: ops.conv3d_g16_mfma(gy, wpb, None, 16, 4, 1, 3, (n, n, n), mask=x, out=dxo), reps)
        print(f"wide bwd-data {n} direct:       {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF")
        ww = ops.pack_wino16_k4(wb)
        for ppc in ((6, 3, 9, 18, 65542, 65539, 65548) if n == 35 else (2, 1, 5, 10)):
            y = ops.conv3d_k4_wino16_bwd(gy, ww, x, ppc=ppc)
            err = float((y - d).abs().max() / d.abs().max())
            us = timeit(lambda: ops.conv3d_k4_wino16_bwd(gy, ww, x, out=dxo, ppc=ppc), reps)
            print(f"wide bwd-data {n} winograd ppc {ppc:2d}: {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF   max|d| / max = {err:.2e}")


def wide_wgrad_main(B, reps):
    """The wide decoder's conv2 / conv1 weight gradients: direct 16-row kernel (+ reduction) against wgrad16_wino.hip."""
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(6)
    for w in (32, 16):
        x = torch.relu(torch.randn(B, 16, w + 3, w + 3, w + 3, generator=g) * 0.7).to(dev)
        gy = (torch.randn(B, 16, w, w, w, generator=g) * (torch.rand(B, 16, w, w, w, generator=g) < 0.6)).to(dev)
        macs = B * 16 * 16 * 64 * w ** 3
        wb = ops.WgradBatch(dev)
        out_d, out_w = torch.empty(16 * 16 * 64, device=dev), torch.empty(16 * 16 * 64, device=dev)

        def direct():
            wb.add(gy, x, 4, 1, 0, 0, out_d)
            wb.finish()
        us = timeit(direct, reps)
        print(f"wide wgrad {w} direct (+ reduce):        {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF")
        for zs in ((1, 2) if w == 32 else (4, 2, 8, 1)):
            def wino():
                base = wb.reserve(256 * 16384 * 4)
                n = ops.wgrad16_k4_wino_partial(gy, x, base, zsplit=zs)
                wb.add_job(base, out_w, n, 16384)
                wb.finish()
            us = timeit(wino, reps)
            err = float((out_w - out_d).abs().max() / out_d.abs().max())
            print(f"wide wgrad {w} winograd zsplit {zs} (+ reduce): {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF   max|d| / max = {err:.2e}")


def wgrad_main(B, reps):
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(2)
    x = torch.relu(torch.randn(B, 8, 35, 35, 35, generator=g) * 0.7).to(dev)
    gy = (torch.randn(B, 8, 32, 32, 32, generator=g) * (torch.rand(B, 8, 32, 32, 32, generator=g) < 0.6)).to(dev)
    macs = B * 8 * 8 * 64 * 32 ** 3
    d = ops.wgrad(gy, x, 4, 1, 0, out_mode=0).reshape(-1)
    us = timeit(lambda: ops.wgrad(gy, x, 4, 1, 0, out_mode=0), reps)
    print(f"wgrad direct  (wgrad_k4_mfma + reduce): {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF algorithmic")
    for zs in (1, 2, 3):
        w = ops.wgrad_k4_wino(gy, x, zsplit=zs).reshape(-1)
        err = float((w - d).abs().max() / d.abs().max())
        us = timeit(lambda: ops.wgrad_k4_wino(gy, x, zsplit=zs), reps)
        print(f"wgrad winograd (zsplit {zs}, + reduce):     {us:7.1f} us  {2 * macs / us / 1e6:6.1f} TF algorithmic   max|d| / max = {err:.2e}")


if __name__ == "__main__":
    if "--fwd" in sys.argv:
        sys.argv.remove("--fwd")
        ap = argparse.ArgumentParser()
        ap.add_argument("--batch", type=int, default=16)
        ap.add_argument("--reps", type=int, default=50)
        a = ap.parse_args()
        fwd_main(a.batch, a.reps)
        sys.exit(0)
    if "--wide" in sys.argv:
        sys.argv.remove("--wide")
        ap = argparse.ArgumentParser()
        ap.add_argument("--batch", type=int, default=16)
        ap.add_argument("--reps", type=int, default=20)
        a = ap.parse_args()
        wide_main(a.batch, a.reps)
        wide_wgrad_main(a.batch, a.reps)
        sys.exit(0)
    if "--wgrad" in sys.argv:
        sys.argv.remove("--wgrad")
        ap = argparse.ArgumentParser()
        ap.add_argument("--batch", type=int, default=16)
        ap.add_argument("--reps", type=int, default=50)
        a = ap.parse_args()
        wgrad_main(a.batch, a.reps)
        sys.exit(0)
    main()
