"""ORACLE -- test infrastructure only (see oracle/nvf_oracle.py for the rules).

CPU restatement of the reference's grid generation (util_get_grids.py:19-46): for every voxel of every level-5
cube the Euclidean distance to the nearest input point; gt_grid = (dist == 0).  The reference asks an open3d
KD-tree (not installable offline: that boundary is "parity unpinned"); the nearest *distance* is unambiguous,
so scipy's cKDTree gives the same numbers.  Octree origins are pinned against the reference's own get_octree
executable (oracle/_ref, tests/golden/octree.npz)."""
import numpy as np
from scipy.spatial import cKDTree


def grids(points, origins):
    pts = np.asarray(points, np.float64)
    origins = np.asarray(origins, np.float64)
    ax = np.arange(32, dtype=np.float64)
    cube = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1)            # [32,32,32,3] = (i, j, k)
    q = (cube[None] + origins[:, None, None, None, :]).reshape(-1, 3)
    d, _ = cKDTree(pts).query(q, k=1)
    dist = d.reshape(len(origins), 1, 32, 32, 32)
    return (dist == 0).astype(np.uint8), dist
