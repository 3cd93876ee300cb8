"""Diagnostic: the wide decoder's slab-reduction jobs and channel-sum tensors of one batch-16 step (what the 49 us launch reads)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nvfpcc_amd import ops

args = bench.parse_args(["--ch", "8", "--chanstr", "16,32,16,16"])
eng = bench.build_engine(args, torch.device("cuda"), 1)
orig = ops.WgradBatch.finish_with_sums


def spy(self, tensors, outs, addends=None, adam=None):
    tot = 0
    for j in self.jobs:
        print(f"job: {j[2]:5d} slabs x {j[3]:7d} floats = {j[2] * j[3] * 4 / 1e6:7.2f} MB")
        tot += j[2] * j[3] * 4
    print(f"slabs total {tot / 1e6:.1f} MB")
    st = 0
    for t in tensors:
        print("sum over", tuple(t.shape), f"{t.numel() * 4 / 1e6:.2f} MB")
        st += t.numel() * 4
    print(f"channel-sum inputs total {st / 1e6:.1f} MB")
    return orig(self, tensors, outs, addends=addends, adam=adam)


ops.WgradBatch.finish_with_sums = spy
import numpy as np
eng.train_step(np.arange(16), 1)
torch.cuda.synchronize()
