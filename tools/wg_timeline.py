"""Timeline of the five-gradient launch (tuning build -DNVF_WG_STAMP, NVF_LIB=...): per job, when its workgroups start and
end relative to the launch's first stamp, and how many workgroups are resident over time."""
import ctypes, re, sys, io, os
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from nvfpcc_amd._lib import lib

args = bench.parse_args([])
args.blocks, args.distinct = 917, 128
eng = bench.build_engine(args, torch.device("cuda"), 1)
rng = np.random.default_rng(0)
for _ in range(3):
    eng.train_step(rng.permutation(917)[:16], 1)
torch.cuda.synchronize()
L = lib()
L.nvf_debug_wg_stamps.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 8192)()
assert L.nvf_debug_wg_stamps(buf, 8192) == 0
st = np.array(buf, dtype=np.uint64).reshape(4096, 2).astype(np.int64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
names = sys.argv[2].split(",") if len(sys.argv) > 2 else None
st = st[:n]
t0 = st[:, 0].min()
s, e = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0      # us (100 MHz)
print(f"launch: first start 0, last end {e.max():.1f} us, {n} workgroups")
if names:
    lo = 0
    for spec in names:
        nm, cnt = spec.split(":")
        cnt = int(cnt)
        if cnt:
            ss, ee = s[lo:lo + cnt], e[lo:lo + cnt]
            print(f"{nm:10s} n={cnt:4d}  start {ss.min():6.1f} .. {ss.max():6.1f}  end {ee.min():6.1f} .. {ee.max():6.1f}  duration mean {np.mean(ee - ss):6.1f} max {np.max(ee - ss):6.1f}")
        lo += cnt
for t in range(0, int(e.max()) + 5, 5):
    print(f"t={t:3d} us: resident {int(((s <= t) & (e > t)).sum())}")
