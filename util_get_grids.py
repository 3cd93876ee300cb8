#!/usr/bin/env python3
"""`python util_get_grids.py cloud.ply 5` of the reference (util_get_grids.py; README step 1b): reads
`{fid}_l5_origins.txt`, writes `{fid}_l5_origins.npy`, `{fid}_l5_gt_grid.npy`, `{fid}_l5_dist.npy` -- distances
from the gfx950 kernel nvf_nearest_dist2 instead of a per-voxel KD-tree loop."""
import sys

import numpy as np

from nvfpcc_amd.preprocess import build_grids, read_ply_xyz

if __name__ == "__main__":
    fid = sys.argv[1].split('/')[-1][:-4]
    lx = int(sys.argv[2]) if len(sys.argv) == 3 else 5
    origins = np.loadtxt(f'{fid}_l{lx}_origins.txt', delimiter=',', ndmin=2)
    np.save(f'{fid}_l{lx}_origins', origins)
    gt, dist = build_grids(read_ply_xyz(sys.argv[1]), origins)
    np.save(f'{fid}_l{lx}_gt_grid', gt)
    np.save(f'{fid}_l{lx}_dist', dist)
