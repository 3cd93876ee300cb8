#!/usr/bin/env python3
"""Times the heads' forward + loss / backward-data as two launches and as the one cooperative launch (batch 16, narrow)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops
from tools.wino_bench import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [(16, 8), (8, 16), (8, 32)]
g = torch.Generator().manual_seed(1)
dev = lambda t: t.cuda()
xs = [dev(torch.randn(B, c, s, s, s, generator=g)) for c, s in shapes]
gts = [dev((torch.rand(B, 1, s, s, s, generator=g) > 0.75).float()) for c, s in shapes]
dist = dev(torch.rand(B, 1, 32, 32, 32, generator=g))
ws = [torch.randn(1, c, 3, 3, 3, generator=g) * 0.1 for c, s in shapes]
bs = [dev(torch.randn(1, generator=g)) for _ in shapes]
packed = [ops.pack_conv_weight(dev(w)) for w in ws]
wf, wb = [p[0] for p in packed], [p[1] for p in packed]
cs = [c for c, s in shapes]
masks = [None, None, xs[2]]
args = ([0.85, 0.85, 0.9], [0.0, 0.0, 1.0], [1, 2, 0])
loss = torch.empty(4, device="cuda")
ctx = ops.StepCtx()


def two():
    ps = ops.heads3_fwd(xs, wf, bs)
    ops.heads3_loss_bwd_data(ps, gts, [None, None, dist], *args, loss, wb, cs, masks)


def one():
    ops.heads3_fwd_loss_bwd_data(xs, wf, bs, gts, [None, None, dist], *args, loss, wb, masks, ctx)


print(f"two launches: {timeit(two, 50):7.1f} us", flush=True)
print(f"one launch:   {timeit(one, 50):7.1f} us", flush=True)
