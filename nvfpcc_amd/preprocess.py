"""Pre-processing of a 10-bit voxelised point cloud into the three `*_l5_*.npy` files the trainer reads
(SURVEY.md section 8, row f2):

  * level-5 octree partition: the 32^3 leaf cubes in the reference's depth-first child order
    (child index = [x >= mid] + 2 [y >= mid] + 4 [z >= mid], get_octree.cpp:354-411, 598-611, 787-795) and the
    breadth-first child-occupancy bit string down to level 5 (:574-595, 797-800);
  * per-cube occupancy and distance grids: dist = Euclidean distance of every voxel of every cube to the
    nearest input point, gt_grid = (dist == 0), axes (x, y, z) (util_get_grids.py:19-46) -- the distances come
    from the gfx950 kernel nvf_nearest_dist2 instead of 30 M KD-tree queries in a Python loop.
"""
import numpy as np
import torch

from ._lib import lib, check

LEAF = 32
ROOT = 1024


def read_ply_xyz(path):
    """Integer x y z of an ASCII PLY (extra per-vertex properties are ignored), get_octree.cpp:751-778."""
    with open(path) as f:
        n = 0
        for line in f:
            if line.startswith("element vertex"):
                n = int(line.split()[2])
            if line.strip() == "end_header":
                break
        pts = np.loadtxt(f, max_rows=n, usecols=(0, 1, 2), ndmin=2)
    return np.asarray(pts, np.int64)


def _child_path_key(cells, levels):
    """Sort key = the sequence of child indices from the root: x is the least significant bit of a level."""
    key = np.zeros(cells.shape[0], np.int64)
    for lv in range(levels):
        bit = levels - 1 - lv
        idx = ((cells[:, 0] >> bit) & 1) | (((cells[:, 1] >> bit) & 1) << 1) | (((cells[:, 2] >> bit) & 1) << 2)
        key = key * 8 + idx
    return key


def octree_level5(points):
    """Returns (origins int64 [N,3] in the reference's traversal order, subtree bit string)."""
    pts = np.asarray(points, np.int64)
    if pts.min() < 0 or pts.max() >= ROOT:
        raise ValueError("coordinates must lie in [0, 1024)")
    cells = np.unique(pts // LEAF, axis=0)                      # level-5 cells, 32 per axis
    order = np.argsort(_child_path_key(cells, 5), kind="stable")
    origins = cells[order] * LEAF
    # breadth-first occupancy: for every node of level 0..5, eight bits for its children
    bits = []
    for level in range(0, 6):
        size = ROOT >> level                                      # node edge at this level
        nodes = np.unique(pts // size, axis=0)
        nodes = nodes[np.argsort(_child_path_key(nodes, level), kind="stable")] if level else nodes
        kids = np.unique(pts // (size // 2), axis=0)
        occupied = set(map(tuple, kids.tolist()))
        for nx, ny, nz in nodes.tolist():
            for i in range(8):
                c = (2 * nx + (i & 1), 2 * ny + ((i >> 1) & 1), 2 * nz + ((i >> 2) & 1))
                bits.append("1" if c in occupied else "0")
    return origins, "".join(bits)


def write_origins_txt(path, origins):
    with open(path, "w") as f:
        for x, y, z in np.asarray(origins, np.int64).tolist():
            f.write(f"{x},{y},{z}\n")


def _neighbour_lists(origins):
    cells = (np.asarray(origins, np.int64) // LEAF)
    index = {tuple(c): i for i, c in enumerate(cells.tolist())}
    off, idx = [0], []
    steps = [(dx, dy, dz) for dx in range(-2, 3) for dy in range(-2, 3) for dz in range(-2, 3)]
    steps.sort(key=lambda s: s[0] * s[0] + s[1] * s[1] + s[2] * s[2])     # nearest blocks first: tight bounds early
    for cx, cy, cz in cells.tolist():
        for dx, dy, dz in steps:
            j = index.get((cx + dx, cy + dy, cz + dz))
            if j is not None:
                idx.append(j)
        off.append(len(idx))
    return np.asarray(off, np.int32), np.asarray(idx, np.int32)


def build_grids(points, origins, device="cuda"):
    """(gt_grid uint8 [N,1,32,32,32], dist float64 [N,1,32,32,32]) of util_get_grids.py:41-46, on the GPU."""
    pts = np.asarray(points, np.int64)
    origins = np.asarray(origins, np.int64)
    n = origins.shape[0]
    cell_of = {tuple(c): i for i, c in enumerate((origins // LEAF).tolist())}
    blk = np.fromiter((cell_of[tuple(c)] for c in (pts // LEAF).tolist()), np.int64, pts.shape[0])
    order = np.argsort(blk, kind="stable")
    spts = pts[order].astype(np.int32)
    blk_off = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(blk, minlength=n), out=blk_off[1:])
    nb_off, nb_idx = _neighbour_lists(origins)
    dev = torch.device(device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_pts, d_off, d_org, d_nbo, d_nbi = t(spts), t(blk_off), t(origins.astype(np.int32)), t(nb_off), t(nb_idx)
    d2 = torch.empty((n, LEAF, LEAF, LEAF), dtype=torch.int32, device=dev)
    check(lib().nvf_nearest_dist2(d_pts.data_ptr(), d_off.data_ptr(), d_org.data_ptr(), d_nbo.data_ptr(),
                                  d_nbi.data_ptr(), d2.data_ptr(), n, torch.cuda.current_stream().cuda_stream),
          "nvf_nearest_dist2")
    d2 = d2.cpu().numpy().astype(np.float64).reshape(n, 1, LEAF, LEAF, LEAF)
    dist = np.sqrt(d2)
    return (dist == 0).astype(np.uint8), dist


def preprocess(ply_path, level=5, device="cuda"):
    """`get_octree` + `util_get_grids.py` in one call; writes the reference's five output files."""
    if level != 5:
        raise NotImplementedError("the codec is built around level-5 (32^3) leaf cubes")
    fid = ply_path.split("/")[-1][:-4]
    pts = read_ply_xyz(ply_path)
    origins, subtree = octree_level5(pts)
    write_origins_txt(f"{fid}_l5_origins.txt", origins)
    with open(f"{fid}_l5_subtree.txt", "w") as f:
        f.write(subtree)
    gt, dist = build_grids(pts, origins, device)
    np.save(f"{fid}_l5_origins", origins.astype(np.float64))       # np.loadtxt gives float64 (util_get_grids.py:16-17)
    np.save(f"{fid}_l5_gt_grid", gt)
    np.save(f"{fid}_l5_dist", dist)
    return origins, gt, dist
