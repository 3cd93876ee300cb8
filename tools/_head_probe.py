import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfpcc_amd import ops
from tools.trunk_bench import timeit
dev = torch.device("cuda")
for C in (8, 16):
    x = torch.randn(16, C, 32, 32, 32, device=dev)
    w = torch.randn(C * 27, device=dev) * 0.1
    b = torch.zeros(1, device=dev)
    y = torch.empty(16, 1, 32, 32, 32, device=dev)
    for act in (2, 77, 78):
        us = timeit(lambda: ops.conv3d_gather(x, w, b, 1, 3, 1, 1, (32, 32, 32), act=act, out=y), reps=50)
        print(f"C={C} act={act}: {us:.1f} us", flush=True)
