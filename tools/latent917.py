import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.parse_args([])
eng = bench.build_engine(args, torch.device("cuda"), 1)
for _ in range(2):
    eng.latent_step(1)
torch.cuda.synchronize()
import time
t=time.perf_counter()
for _ in range(5):
    eng.latent_step(1)
torch.cuda.synchronize()
print("latent step ms", (time.perf_counter()-t)/5*1e3)
