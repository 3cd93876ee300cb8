"""Synthetic leaf blocks standing in for ``longdress_vox10_1300`` (not available offline).

Each block is a 32^3 occupancy grid holding a thin random quadric sheet at
2.5-3.5 % occupancy plus the exact Euclidean distance of every voxel to the
nearest occupied voxel of the same block -- the semantics of
/root/reference/util_get_grids.py:36-46 (``dist`` = nearest-point distance,
``gt_grid = dist == 0``).  Array dtypes/shapes match what
``LoadedVoxelDataset`` reads (/root/reference/utils/dataloader.py:155-157).
"""
import numpy as np
from scipy import ndimage

GRID = 32


def make_block(i, base_seed=20221):
    """Return (gt u8 [1,32,32,32], dist f64 [1,32,32,32]) for block ``i``."""
    rng = np.random.default_rng(base_seed + int(i))
    ax = (np.arange(GRID, dtype=np.float64) + 0.5) / GRID * 2.0 - 1.0
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    n = rng.normal(size=3)
    n /= np.linalg.norm(n)
    a = rng.normal(scale=0.35, size=(3, 3))
    a = 0.5 * (a + a.T)
    d = rng.uniform(-0.3, 0.3)
    p = np.stack([z, y, x], 0)
    q = np.tensordot(n, p, 1) + d + np.einsum("i...,ij,j...->...", p, a, p)
    grad = n[:, None, None, None] + 2.0 * np.tensordot(a, p, 1)
    sdf = np.abs(q) / (np.linalg.norm(grad, axis=0) + 1e-9)
    k = int(rng.integers(820, 1148))  # 2.5 % .. 3.5 % of 32768
    thr = np.partition(sdf.reshape(-1), k - 1)[k - 1]
    gt = sdf <= thr
    dist = ndimage.distance_transform_edt(~gt)
    return gt.astype(np.uint8)[None], dist.astype(np.float64)[None]


def make_blocks(n, base_seed=20221, start=0):
    gts = np.empty((n, 1, GRID, GRID, GRID), np.uint8)
    dists = np.empty((n, 1, GRID, GRID, GRID), np.float64)
    for j in range(n):
        gts[j], dists[j] = make_block(start + j, base_seed)
    return gts, dists


def make_origins(n):
    """Distinct int cube origins on the 32-lattice of a 1024^3 volume."""
    idx = np.arange(n)
    return np.stack([(idx // 1024) % 32, (idx // 32) % 32, idx % 32], 1).astype(np.float64) * GRID


def write_dataset(prefix, n, base_seed=20221):
    """Write the three ``*_l5_*.npy`` files ``NVFPCC.py train`` expects."""
    gts, dists = make_blocks(n, base_seed)
    np.save(f"{prefix}_l5_origins.npy", make_origins(n))
    np.save(f"{prefix}_l5_gt_grid.npy", gts)
    np.save(f"{prefix}_l5_dist.npy", dists)
    return gts, dists
